"""The per-window tail (f_movie, IEF, SMPL records) beside a two-part ResNet on the engine's priority streams, 300 times, against the tail run
alone: every record must come out the same.  Round 5 found smpl_pose_kernel compiled with SLP-packed fp32 (v_pk_*_f32) failing this in 20-60 %
of the launches (wrong A[j][1][3] for joints 16-23 of odd instances: the last quarter of a wave); the build now passes -fno-slp-vectorize.
    [DBG_DT=bf16|f16x3|f32] python tools/tail_race_check.py"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import Config
from human_dynamics_amd import assets, _lib as L
from human_dynamics_amd.evaluation.tester import Tester, OUTPUT_KEYS
from human_dynamics_amd import dist as hd
w, s = assets.make_synthetic_weights(0), assets.make_synthetic_smpl(2)
import os
t = Tester(Config(batch_size=8), weights=w, smpl=s, dtype=os.environ.get("DBG_DT", "bf16"), device="cuda:0")
eng = t.engine
dev = eng.device
if os.environ.get("DBG_BLEND"):          # hmmr_debug_t.smpl_blend_mfma: 2 = the packed-FMA vector form (hand-written v_pk_fma_f32), 1 = exact-fp32 MFMA
    from human_dynamics_amd import engine as E
    E.set_debug(smpl_blend_mfma=int(os.environ["DBG_BLEND"]))
n = 128
g = torch.Generator().manual_seed(1)
windows = torch.randn((16, 20, 2048), generator=g).to(dev)
layout, rec_len = hd.record_layout(2)
ref = torch.zeros((n, rec_len), device=dev)
t.predict_strips_records(windows, n, out=ref)
om_ref = t.predict_strips_omegas(windows, n).clone()
torch.cuda.synchronize()
ws_ref = eng._ws["smpl"].buf.clone()
LDF, LDA = 224, 288
mp = 384
featA = lambda b: (b[:mp * LDF * 4].view(torch.float32).reshape(mp, LDF), b[mp * LDF * 4:mp * LDF * 4 + mp * LDA * 4].view(torch.float32).reshape(mp, LDA))
frames = torch.rand((128, 224, 224, 3), device=dev) * 2 - 1
phi = torch.empty((128, 2048), device=dev)
s_tail = torch.cuda.Stream(priority=int(os.environ.get("DBG_TAIL_PRIORITY", "0")))
REPS = int(os.environ.get("DBG_REPS", "300"))
off = {k: (o, sz) for k, shp, o, sz in layout}
# DBG_MODE: "with 2-part resnet on priority streams" (default; HMMR_RESNET_PRIORITY sets the side streams' priority, 0 = none) | "alone" |
# "smpl only beside the resnet"
for mode in (os.environ.get("DBG_MODE", "with 2-part resnet on priority streams"),):
    bad = 0
    for rep in range(REPS):
        rec = torch.full((n, rec_len), float("nan"), device=dev)
        torch.cuda.synchronize()
        cur = torch.cuda.current_stream()
        if mode != "alone":
            for i, (a, b) in enumerate(((0, 64), (64, 128))):
                sc = eng.side_stream(i)
                sc.wait_stream(cur)
                with torch.cuda.stream(sc):
                    eng.resnet(frames[a:b], out=phi[a:b], parts=1, ws_key="resnet%d" % i)
        with torch.cuda.stream(s_tail):
            s_tail.wait_stream(cur)
            if mode.startswith("smpl only"):
                t.records_from_omegas(om_ref, rec)
            else:
                om = t.predict_strips_omegas(windows, n)
                om_at_launch = om.clone()                       # (a copy kernel on s_tail between the IEF and the SMPL kernels)
                t.records_from_omegas(om, rec)
        torch.cuda.synchronize()
        if not mode.startswith("smpl only") and (not torch.equal(om, om_ref) or not torch.equal(om_at_launch, om_ref)):
            dd = (om_at_launch != om_ref)
            print("   om differs: final", int((om != om_ref).sum()), "at-launch copy", int(dd.sum()), "rows", dd.any(2).nonzero().tolist()[:6], "cols", dd.any(0).any(0).nonzero().flatten().tolist()[:12])
        if not torch.equal(rec, ref):
            bad += 1
            d = (rec != ref)
            fr = d.any(1).nonzero().flatten().tolist()
            ks = [k for k, (o, sz) in off.items() if bool(d[:, o:o + sz].any())]
            f_r, a_r = featA(ws_ref)
            f_g, a_g = featA(eng._ws["smpl"].buf)
            df, da = (f_g != f_r), (a_g != a_r)
            if bad <= 8:
                print("  frames", fr[:6], "| feat rows differing", df.any(1).nonzero().flatten().tolist()[:6], "cols", df.any(0).nonzero().flatten().tolist()[:8],
                      "| A rows differing", da.any(1).nonzero().flatten().tolist()[:6], "cols", da.any(0).nonzero().flatten().tolist()[:16])
            if os.environ.get("DBG_DUMP") and bad <= 6:
                # the first wrong joint of each wrong instance: got / expected A[j][1][3], the error, and the candidates it might be made of
                # (the parent's A row 1, this joint's A rows: everything the chain step G_i = G_p [R_i | t_i] touches)
                PAR = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]
                for r_ in da.any(1).nonzero().flatten().tolist()[:3]:
                    j_ = min(c // 12 for c in da[r_].nonzero().flatten().tolist())
                    g_, e_ = a_g[r_].reshape(24, 12), a_r[r_].reshape(24, 12)
                    print("    instance", r_, "first wrong joint", j_, "parent", PAR[j_], "| got", float(g_[j_, 7]), "expected", float(e_[j_, 7]),
                          "err", float(g_[j_, 7] - e_[j_, 7]))

                    def chain(inst):                             # float64 forward kinematics of one instance: the global (G) and local (Lc) transforms [24][12]
                        o_ = om_ref.reshape(-1, 85)[inst].double().cpu().numpy()
                        th, be = o_[3:75].reshape(24, 3), o_[75:85]
                        vs = s["v_template"].astype(np.float64) + (be @ s["shapedirs"].astype(np.float64)).reshape(6890, 3)
                        J = s["J_regressor"].astype(np.float64).T @ vs
                        G, Lc = np.zeros((24, 12)), np.zeros((24, 12))
                        for q in range(24):
                            e3 = th[q] + 1e-8
                            an = np.sqrt((e3 * e3).sum()); rr = th[q] / an; c_, s_ = np.cos(an), np.sin(an)
                            K = np.array([[0, -rr[2], rr[1]], [rr[2], 0, -rr[0]], [-rr[1], rr[0], 0]])
                            Lc[q, :9] = (c_ * np.eye(3) + (1 - c_) * np.outer(rr, rr) + s_ * K).reshape(9)
                            Lc[q, 9:] = J[q] - (J[PAR[q]] if q else 0)
                            if q == 0:
                                G[0] = Lc[0]
                            else:
                                Gp = G[PAR[q]]
                                G[q, :9] = (Gp[:9].reshape(3, 3) @ Lc[q, :9].reshape(3, 3)).reshape(9)
                                G[q, 9:] = Gp[:9].reshape(3, 3) @ Lc[q, 9:] + Gp[9:]
                        return G, Lc
                    G, Lc = chain(r_)
                    G2, Lc2 = chain(r_ - 1)
                    p_ = PAR[j_]
                    for nm, GG, LL in (("this", G, Lc), ("even", G2, Lc2)):
                        print("      %s: G_p[3..5] %s G_p[9..11] %s Lc[9..11] %s | products row1 %s | o[9..11] %s" % (
                            nm, np.round(GG[p_, 3:6], 5).tolist(), np.round(GG[p_, 9:12], 5).tolist(), np.round(LL[j_, 9:12], 5).tolist(),
                            np.round(GG[p_, 3:6] * LL[j_, 9:12], 5).tolist(), np.round(GG[j_, 9:12], 5).tolist()))
                    print("      Lc rot", np.round(Lc[j_, :9], 4).tolist(), "G_p rot", np.round(G[p_, :9], 4).tolist(), "G rot", np.round(G[j_, :9], 4).tolist())
            if bad <= 0:
                print(mode, "rep", rep, "frames", fr[:10], "fields", ks, "nan", int(torch.isnan(rec).sum()))
    print(os.environ.get("DBG_DT", "bf16"), {k: v for k, v in os.environ.items() if k.startswith("HMMR_")}, mode, "| tail priority", os.environ.get("DBG_TAIL_PRIORITY", "0"), "| blend form", os.environ.get("DBG_BLEND", "default"), "| library", os.path.basename(L.LIB_PATH), ": bad", bad, "of", REPS)
