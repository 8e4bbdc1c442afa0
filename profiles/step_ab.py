"""Same-box A/B of the whole inference step: runs bench.py's plain step (no extra legs) in a child process per library build.
usage: python3 profiles/step_ab.py <libA.so|default> <libB.so|default> [rounds]   — alternates A, B, A, B ..."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = (
    "import os, sys, runpy\n"
    "sys.path.insert(0, %r)\n"
    "from sapcu_amd import _lib\n"
    "p = os.environ.get('SAPCU_AB_LIB')\n"
    "if p: _lib.LIB_PATH = p\n"
    "sys.argv = ['bench.py', '--steps', '8', '--warmup', '2', '--no-cpu-baseline', '--no-strong-leg', '--no-m100', '--no-ref-default', '--no-roofline']\n"
    "runpy.run_path(%r, run_name='__main__')\n" % (ROOT, os.path.join(ROOT, "bench.py")))


def run(lib):
    env = dict(os.environ)
    env.pop("SAPCU_AB_LIB", None)
    if lib != "default":
        env["SAPCU_AB_LIB"] = os.path.abspath(lib)
    out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)["ms_per_step"]


if __name__ == "__main__":
    a, b = sys.argv[1], sys.argv[2]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    for _ in range(rounds):
        print("A %s: %.3f ms/step" % (a, run(a)), flush=True)
        print("B %s: %.3f ms/step" % (b, run(b)), flush=True)
