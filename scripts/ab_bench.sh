#!/bin/bash
# quick A/B on the GPU box: the bench with and without an env switch, 3 runs each, best step time
# usage: ABVAR=NAME scripts/ab_bench.sh
python - <<'PY'
import json,subprocess,os,sys
def run(env, reps=3):
    best=None
    for _ in range(reps):
        e=dict(os.environ, **env)
        out=subprocess.run([sys.executable,"bench.py","--steps","80","--no-cpu-baseline","--no-host-legs"],capture_output=True,text=True,env=e)
        try:
            d=json.loads(out.stdout.strip().splitlines()[-1])
        except Exception:
            print(out.stdout[-2000:], out.stderr[-3000:]); raise
        if best is None or d["ms_per_step"]<best["ms_per_step"]: best=d
    k=best["kernel_ms"]
    print(env, f"best of {reps}: {best['value']:.0f} windows/s, {best['ms_per_step']:.3f} ms/step;", {a:round(b,3) for a,b in k.items()}, flush=True)
v=os.environ.get("ABVAR","ISV_LEGACY_VISUAL")
run({v:"1"}); run({})
PY
