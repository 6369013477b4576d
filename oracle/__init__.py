"""CPU oracle for the prompt-tts hot path.  TEST INFRASTRUCTURE ONLY.

Plain-PyTorch fp32 (CPU) restatement of the reference's algorithm for the path
named by BASELINE.json:north_star.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import this package; the product
package ``prompt_tts_amd`` never does (tests/test_no_oracle_in_product.py
enforces it).

Pinning status (see DESIGN.md "Oracle"):
  * reference-owned wiring (tts/models.py, tts/ldm/*.py): PINNED against the
    unmodified reference modules imported in the authoring container
    (tests/golden/make_golden.py), outputs committed under tests/golden/.
  * third-party arithmetic (diffusers 0.15.x BasicTransformerBlock / Attention /
    GEGLU / Timesteps / TimestepEmbedding / DDPMScheduler.add_noise, encodec
    0.1.1 decoder): the packages are absent from /root/reference and from this
    image -> "parity unpinned" by the reference itself; restated from the
    published algorithm and cross-checked against torch.nn primitives and the
    locally installed transformers.models.encodec architecture.
"""
