#!/bin/bash
# quick A/B on the GPU box: solver-path tests, then the bench with and without an env switch
# usage: scripts/ab_bench.sh ENVVAR   (runs `ENVVAR=1 python bench.py ...` then plain)
V=${1:-ISV_LEGACY_VISUAL}
python - <<'PY'
import json,subprocess,os,sys
def run(env):
    e=dict(os.environ, **env)
    out=subprocess.run([sys.executable,"bench.py","--steps","60","--no-cpu-baseline","--no-host-legs"],capture_output=True,text=True,env=e)
    try:
        d=json.loads(out.stdout.strip().splitlines()[-1])
    except Exception:
        print(out.stdout[-2000:], out.stderr[-3000:]); raise
    k=d["kernel_ms"]
    print(env, f"{d['value']:.0f} windows/s, {d['ms_per_step']:.3f} ms/step;", {a:round(b,3) for a,b in k.items()})
v=os.environ.get("ABVAR","ISV_LEGACY_VISUAL")
run({v:"1"}); run({})
PY
