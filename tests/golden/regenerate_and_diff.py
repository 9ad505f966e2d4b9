#!/usr/bin/env python
"""Re-runs every golden generator against /root/reference (build container only: the reference does not travel) into a scratch
directory and asserts that the COMMITTED fixtures are reproduced -- so a generator that drifted away from its fixture (or a
fixture edited by hand) does not go unnoticed.  Arrays must agree to 1e-6 relative (the generators run torch CPU kernels whose
thread partitioning may differ from run to run in the last bits); key lists and names must be identical.

Usage (repo root):  python tests/golden/regenerate_and_diff.py          exit code 0 = every fixture reproduced
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
# generator -> the fixtures it writes next to itself
GENERATORS = [
    ("make_sepconv_golden.py", ["sepconv_kat.npz"]),
    ("make_warp_golden.py", ["warp.npz"]),
    ("make_model_goldens.py", ["models.npz", "models_state_dict_keys.json"]),
    ("make_step_goldens.py", ["steps.npz", "steps_names.json"]),
    ("make_bf16_step_golden.py", ["steps_bf16.npz", "steps_bf16_names.json"]),
    ("make_sff_chain_golden.py", ["sff_chain.npz"]),
]
RTOL = 1e-6


def compare(name, old, new):
    bad = []
    if name.endswith(".json"):
        if json.load(open(old)) != json.load(open(new)):
            bad.append("%s: JSON differs" % name)
        return bad
    a, b = np.load(old), np.load(new)
    if sorted(a.files) != sorted(b.files):
        bad.append("%s: keys differ: %s" % (name, sorted(set(a.files) ^ set(b.files))))
        return bad
    for k in a.files:
        x, y = np.asarray(a[k]), np.asarray(b[k])
        if x.shape != y.shape or x.dtype != y.dtype:
            bad.append("%s[%s]: shape/dtype %s %s vs %s %s" % (name, k, x.shape, x.dtype, y.shape, y.dtype))
            continue
        if x.dtype.kind in "fc":
            if k.endswith("_cond") or "_vs_fp32_" in k:
                continue        # fp32-vs-fp64 noise measurements: themselves noise-sized, compared through the tests' tolerances only
            scale = max(float(np.abs(x).max()), 1e-30)
            err = float(np.abs(x.astype(np.float64) - y.astype(np.float64)).max()) / scale
            if err > RTOL:
                bad.append("%s[%s]: max deviation %.3e of its largest element" % (name, k, err))
        elif not np.array_equal(x, y):
            bad.append("%s[%s]: values differ" % (name, k))
    return bad


def main():
    if not os.path.isdir("/root/reference"):
        raise SystemExit("needs /root/reference (build container only)")
    scratch = tempfile.mkdtemp(prefix="golden_regen_")
    problems = []
    try:
        work = os.path.join(scratch, "golden")
        shutil.copytree(HERE, work)                      # generators write next to themselves: run the copies
        # the copies must still find tests/ (weight_recipe, sepconv_cases) and the repo (oracle/)
        env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(REPO, "tests"), REPO, os.path.join(REPO, "sstem-restoration_amd")]))
        for gen, outputs in GENERATORS:
            for o in outputs:
                os.remove(os.path.join(work, o))
            r = subprocess.run([sys.executable, os.path.join(work, gen)], cwd=REPO, env=env, capture_output=True, text=True)
            if r.returncode != 0:
                problems.append("%s failed:\n%s" % (gen, (r.stdout + r.stderr)[-2000:]))
                continue
            for o in outputs:
                if not os.path.exists(os.path.join(work, o)):
                    problems.append("%s did not write %s" % (gen, o))
                else:
                    problems += compare(o, os.path.join(HERE, o), os.path.join(work, o))
            print("%-28s %s" % (gen, "ok" if not any(p.startswith(tuple(outputs)) for p in problems) else "DIFFERS"), flush=True)
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
    if problems:
        print("\n".join(problems))
        raise SystemExit(1)
    print("every committed fixture is reproduced by its generator")


if __name__ == "__main__":
    main()
