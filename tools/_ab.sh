for t in 0 512 256 128 0; do
  export PT_GEMM_TILE=$t
  timeout -k 10 200 python tools/decode_probe.py f32 > gpurun_out/tile_$t.log 2>&1
  echo "tile $t: $(grep -E 'pt_gemm' gpurun_out/tile_$t.log | awk '{print $(NF-1)}' | tr '\n' ' ') | $(grep 'decode 64' gpurun_out/tile_$t.log)"
done
