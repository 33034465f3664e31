"""Developer tool: repeat the same solve many times on one context (reset + run) and require bit-identical results: the two-stream
S assembly, the level-scheduled Cholesky, the polled host hand-off and the deterministic reductions must not depend on timing.
Needs a GPU:  python tools/soak_determinism.py [config | C | T] [repeats] [max LM iterations]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401

from xrsfm_amd import capi, synth


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "L"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    # "C": a dissected photo collection (2400 photos in clusters): level schedule with the level look-ahead on two streams (round 6)
    full = synth.make_collection(n_cams=2400, n_points=100000, seed=4, cams_per_cluster=60) if cfg == "C" else \
        (synth.make_collection(**synth.CONFIGS[cfg]) if cfg == "T" else synth.make_problem(**synth.CONFIGS[cfg]))
    prob = capi.ProblemArrays(**{k: np.array(full[k], copy=True) for k in capi.ProblemArrays.FIELDS})
    ctx = capi.Context(prob)
    opt = capi.default_options(max_iterations=int(sys.argv[3])) if len(sys.argv) > 3 else capi.default_options()
    ref = None
    for i in range(n):
        ctx.reset()
        s = ctx.run(opt)
        q, t, P = ctx.download()
        key = (s.final_cost, s.n_successful, s.n_unsuccessful, float(q.sum()), float(t.sum()), float(P.sum()))
        if ref is None:
            ref = key
        assert key == ref, f"run {i} differs: {key} vs {ref}"
    print(f"{cfg}: {n} runs bit-identical, final cost {ref[0]!r}, {ref[1]}+{ref[2]} steps")
    ctx.close()


if __name__ == "__main__":
    main()
