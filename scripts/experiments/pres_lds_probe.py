"""GPU probe of the LDS-transform form of Pres_2::exec (csrc/pres_lds.h) against the staged rocFFT form: largest differences of
p / ut / vt / wt and the time of both forms and of the three stages. python scripts/experiments/pres_lds_probe.py [case:itot:jtot:ktot ...]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from microhh_amd.model import HotPath

def timed(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

specs = sys.argv[1:] or ["drycblles:64:64:32", "drycblles:256:256:256", "drycblles:512:512:512", "gabls1:1024:1024:256"]
for spec in specs:
    case, it, jt, kt = spec.split(":")
    kw = {"dtype": np.float32} if case == "gabls1" else {}
    hp = HotPath(case, int(it), int(jt), int(kt), device="cuda:0", dt=0.5, **kw)
    assert hp.lib.mhh_pres_plan_has_lds_form(hp.plan) == 1
    hp.cyclic_prognostic(); hp.exec_viscosity(); hp.rhs(); hp.sync()
    keep = [t.clone() for t in (hp.ut, hp.vt, hp.wt)]
    out = {}
    for form in ("staged", "lds"):
        for t, k in zip((hp.ut, hp.vt, hp.wt), keep): t.copy_(k)
        hp.p.zero_()
        os.environ["MHH_PRES_LDS"] = "0" if form == "staged" else "1"
        hp.pres(); hp.sync()
        out[form] = [t.clone() for t in (hp.p, hp.ut, hp.vt, hp.wt)]
    diffs = []
    for a, b in zip(out["staged"], out["lds"]):
        diffs.append(float((a - b).abs().max()) / max(float(a.abs().max()), 1e-300))
    del out
    res = {}
    for form in ("staged", "lds"):
        os.environ["MHH_PRES_LDS"] = "0" if form == "staged" else "1"
        res[form] = timed(hp.pres)
    st = [timed(lambda s=s: hp._ok(hp.lib.mhh_pres_lds_stage(hp.plan, hp.G, C.byref(hp.fields), hp.dt, s, hp.stream))) for s in (1, 2, 3)]
    print(f"{spec:28s} rel diff p/ut/vt/wt " + " ".join(f"{d:.1e}" for d in diffs) + f" | staged {res['staged']:.3f} ms  lds {res['lds']:.3f} ms  stages " + " ".join(f"{x:.3f}" for x in st), flush=True)
    hp.close(); del hp, keep
    torch.cuda.empty_cache()
