#!/usr/bin/env python
"""Upper bounds for training-step changes: the headline step with chosen ops replaced by no-ops (results are wrong; the step's
time is what is read), or -- prefix 2x -- launched twice (results stay finite: the honest form, a skipped producer leaves garbage
and the MFMA kernels downstream run faster on it).  python tools/step_skip_probe.py "" 2xgroupnorm_bwd wgrad_group -> ms per step.
What a kernel costs IN the step (two streams filling each other's gaps) is not its time alone; this measures the former."""
import json
import os
import runpy
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("_PT_PROBE_CHILD"):
    from prompt_tts_amd import ops
    for name in [n for n in os.environ.get("PT_PROBE_SKIP", "").split(",") if n]:
        twice = name.startswith("2x")
        name = name[2:] if twice else name
        assert hasattr(ops, name), name
        if twice:       # launched twice: the data stay valid (a skipped producer leaves garbage, and MFMA kernels run faster on it)
            def both(*a, _f=getattr(ops, name), **k):
                _f(*a, **k)
                return _f(*a, **k)
            setattr(ops, name, both)
        else:
            setattr(ops, name, lambda *a, **k: None)
    sys.argv = ["bench.py", "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-decode", "--no-replay"]
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
else:
    for skip in sys.argv[1:] or [""]:
        env = dict(os.environ, _PT_PROBE_CHILD="1", PT_PROBE_SKIP=skip)
        out = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, cwd=ROOT)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        ms = json.loads(line[-1])["ms_per_step"] if line else None
        print(f"skip [{skip}]: {ms if ms is None else round(ms, 2)} ms per step" + ("" if line else "  " + out.stderr[-300:]), flush=True)
