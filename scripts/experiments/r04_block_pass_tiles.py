"""Round-4 experiment: Schur block pass with the operands of a workgroup's 16 blocks staged in LDS (block_pass_tiles_body).
The tile lists -- per workgroup: stages; per stage: unique a-edges / b-edges / landmarks; per product: packed LDS slots -- are built HERE
with numpy from the library's own block order and product lists (debug hooks) and handed in; the reduced matrix must come out
bit-identical, and linearise + Schur is timed with cuba_hip_time_kernels before and after."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from cuba_amd.capi import HipSolver, load_library
from cuba_amd.graph import flatten
from cuba_amd.synth import synth_named
RK = ((1, float(np.sqrt(5.991))), (1, float(np.sqrt(7.815))))
LDS = 48 * 1024
_ip = C.POINTER(C.c_int32)

def get_ints(h, name):
    n = C.c_size_t()
    assert h.lib.cuba_hip_debug_get_ints(h.h, name.encode(), None, C.byref(n)) == 0
    out = np.zeros(n.value, np.int32)
    assert h.lib.cuba_hip_debug_get_ints(h.h, name.encode(), out.ctypes.data_as(_ip), C.byref(n)) == 0
    return out

def build_tiles(od, pbeg, pend, ea, eb, lm, lim_a=2047, lim_b=2047, lim_l=1023, lim_p=1 << 30, lds=LDS):
    nWG = (len(od) + 15) // 16
    odp = np.full(nWG * 16, -1, np.int64); odp[:len(od)] = od
    stage_ptr = [0]; a_beg = [0]; b_beg = [0]; l_beg = [0]; p_end = []
    TA, TB, TL = [], [], []
    slots = np.zeros(len(ea), np.uint32)
    for wg in range(nWG):
        blocks = odp[16 * wg:16 * wg + 16]
        rng = [(int(pbeg[b]), int(pend[b])) if b >= 0 else (0, 0) for b in blocks]
        pidx = np.concatenate([np.arange(b0, b1) for b0, b1 in rng]) if any(b1 > b0 for b0, b1 in rng) else np.zeros(0, np.int64)
        if len(pidx) == 0:
            stage_ptr.append(stage_ptr[-1]); continue
        pea, peb, plm = ea[pidx], eb[pidx], lm[pidx]
        ua, ub, ul = np.unique(pea), np.unique(peb), np.unique(plm)
        # landmark of every unique edge: an edge belongs to one landmark, and edge ids grow with the landmark
        la = plm[np.searchsorted(np.sort(pea), ua)] if False else None
        oa = np.argsort(pea, kind="stable"); la = plm[oa][np.searchsorted(pea[oa], ua)]
        ob = np.argsort(peb, kind="stable"); lb = plm[ob][np.searchsorted(peb[ob], ub)]
        ca = np.searchsorted(la, ul, side="right"); cb = np.searchsorted(lb, ul, side="right")     # unique a / b edges with landmark <= ul[i]
        cp = np.searchsorted(np.sort(plm), ul, side="right")                                       # products of the workgroup with landmark <= ul[i]
        # greedy stage cuts over the landmarks
        cuts = []; i0 = 0
        while i0 < len(ul):
            a0 = ca[i0 - 1] if i0 else 0; b0 = cb[i0 - 1] if i0 else 0; p0 = cp[i0 - 1] if i0 else 0
            na = ca[i0:] - a0; nb = cb[i0:] - b0; nl = np.arange(1, len(ul) - i0 + 1); npr = cp[i0:] - p0
            ok = (32 * (na + nb) + 48 * nl <= lds) & (na <= lim_a) & (nb <= lim_b) & (nl <= lim_l) & (npr <= lim_p)
            k = int(np.argmin(ok)) if not ok.all() else len(ok)
            assert k >= 1, "one landmark does not fit a stage"
            i0 += k; cuts.append(i0)
        prev = 0
        for c in cuts:
            a0 = ca[prev - 1] if prev else 0; b0 = cb[prev - 1] if prev else 0
            a1, b1 = ca[c - 1], cb[c - 1]
            TA.append(ua[a0:a1]); TB.append(ub[b0:b1]); TL.append(ul[prev:c])
            a_beg.append(a_beg[-1] + (a1 - a0)); b_beg.append(b_beg[-1] + (b1 - b0)); l_beg.append(l_beg[-1] + (c - prev))
            lim = ul[c - 1]
            for (q0, q1) in rng:
                p_end.append(q0 + int(np.searchsorted(lm[q0:q1], lim, side="right")) if q1 > q0 else 0)
            # slots of the products of this stage
            m = (plm >= ul[prev]) & (plm <= lim)
            sa = np.searchsorted(ua, pea[m]) - a0; sb = np.searchsorted(ub, peb[m]) - b0; sl = np.searchsorted(ul, plm[m]) - prev
            slots[pidx[m]] = (sa | (sb << 11) | (sl << 22)).astype(np.uint32)
            prev = c
        stage_ptr.append(stage_ptr[-1] + len(cuts))
    i32 = lambda x: np.ascontiguousarray(np.concatenate(x) if isinstance(x, list) and len(x) and isinstance(x[0], np.ndarray) else np.array(x), dtype=np.int32)
    return dict(nWG=nWG, nStages=stage_ptr[-1], stage_ptr=i32(stage_ptr), a_beg=i32(a_beg), b_beg=i32(b_beg), l_beg=i32(l_beg), p_end=i32(p_end),
                ta=i32(TA), tb=i32(TB), tl=i32(TL), slots=slots)

def set_tiles(h, t, merged, inv8):
    if t is None:
        return h.lib.cuba_hip_debug_set_tiles(h.h, 0, 0, None, None, None, None, None, None, 0, None, 0, None, 0, None, 0, 1, inv8)
    p = lambda a: a.ctypes.data_as(_ip)
    rc = h.lib.cuba_hip_debug_set_tiles(h.h, t["nWG"], t["nStages"], p(t["stage_ptr"]), p(t["a_beg"]), p(t["b_beg"]), p(t["l_beg"]), p(t["p_end"]),
                                        p(t["ta"]), C.c_size_t(len(t["ta"])), p(t["tb"]), C.c_size_t(len(t["tb"])), p(t["tl"]), C.c_size_t(len(t["tl"])),
                                        t["slots"].ctypes.data_as(C.POINTER(C.c_uint32)), C.c_size_t(len(t["slots"])), merged, inv8)
    assert rc == 0, rc

for shape in sys.argv[1:] or ["kitti00"]:
    fp = flatten(synth_named(shape))
    h = HipSolver(fp, RK)
    lib = h.lib
    lib.cuba_hip_debug_get_ints.argtypes = [C.c_void_p, C.c_char_p, _ip, C.POINTER(C.c_size_t)]
    lib.cuba_hip_debug_set_tiles.argtypes = [C.c_void_p, C.c_int, C.c_int, _ip, _ip, _ip, _ip, _ip, _ip, C.c_size_t, _ip, C.c_size_t, _ip, C.c_size_t,
                                             C.POINTER(C.c_uint32), C.c_size_t, C.c_int, C.c_int]
    h.build_structure()
    md = h.max_diagonal(); h.set_lambda(1e-5 * md)
    h.schur(); ref = h.array("hsc"); ref_bsc = h.array("bsc")
    t0 = h.time_kernels(20)["linearize_schur"]
    od, pbeg, pend = get_ints(h, "od_blocks"), get_ints(h, "prod_beg"), get_ints(h, "prod_end")
    ea, eb, lm = get_ints(h, "prod_ea"), get_ints(h, "prod_eb"), get_ints(h, "prod_lm")
    tb = time.time(); tiles = build_tiles(od, pbeg, pend, ea.astype(np.int64), eb.astype(np.int64), lm.astype(np.int64)); tb = time.time() - tb
    staged = len(tiles["ta"]) + len(tiles["tb"]) + len(tiles["tl"])
    print(f"{shape}: {len(ea)} products in {int((od >= 0).sum())} blocks, {tiles['nWG']} workgroups, {tiles['nStages']} stages; staged operands {staged} "
          f"(a {len(tiles['ta'])}, b {len(tiles['tb'])}, landmarks {len(tiles['tl'])}) against {3 * len(ea)} gathers; lists built in {tb:.1f} s (numpy)", flush=True)
    print(f"{shape}: default pass            linearise + Schur {t0 * 1e3:7.2f} us", flush=True)
    if os.environ.get("TILES_CHECK"):
        t = tiles; bad = 0
        for wg in range(t["nWG"]):
            blks = [int(b) for b in od[16 * wg:16 * wg + 16]] + [-1] * 16
            p = [int(pbeg[b]) if b >= 0 else 0 for b in blks[:16]]
            for s in range(t["stage_ptr"][wg], t["stage_ptr"][wg + 1]):
                for gi in range(16):
                    pe = int(t["p_end"][s * 16 + gi])
                    if pe > p[gi]:
                        w = t["slots"][p[gi]:pe].astype(np.int64)
                        sa, sb, sl = w & 2047, (w >> 11) & 2047, w >> 22
                        ok = np.array_equal(t["ta"][t["a_beg"][s] + sa], ea[p[gi]:pe]) and np.array_equal(t["tb"][t["b_beg"][s] + sb], eb[p[gi]:pe]) and np.array_equal(t["tl"][t["l_beg"][s] + sl], lm[p[gi]:pe])
                        bad += not ok
                    p[gi] = max(p[gi], pe)
            for gi in range(16):
                if blks[gi] >= 0 and p[gi] != int(pend[blks[gi]]): bad += 1
        print(f"{shape}: list integrity check: {bad} bad (stage, group) entries", flush=True)
    rp, ci = h.hsc_structure()
    offd = np.ones(len(ci), bool); offd[rp[:-1]] = False          # (the lower triangles of the diagonal blocks are never written: compare the rest)
    tiles2 = build_tiles(od, pbeg, pend, ea.astype(np.int64), eb.astype(np.int64), lm.astype(np.int64), 320, 320, 128, 1024, 1 << 30)
    print(f"{shape}: pipelined variant: {tiles2['nStages']} stages (at most 320 + 320 records, 128 inverses, 1024 products each)", flush=True)
    if os.environ.get("TILES_TRUNCATE"):
        # timing-only hypothesis test: is the pass bound by its LONGEST product list (27 dependent trips of 16 lanes at KITTI-00)?  Every
        # group stops after `cap` products of its block (wrong reduced system, same launch shape)
        cap = int(os.environ["TILES_TRUNCATE"])
        odp = np.full(tiles["nWG"] * 16, -1, np.int64); odp[:len(od)] = od
        for t_ in (tiles, tiles2):
            lim = np.where(odp >= 0, pbeg[np.maximum(odp, 0)] + cap, 0)
            pe = t_["p_end"].reshape(-1, 16)
            wg_of_stage = np.repeat(np.arange(t_["nWG"]), np.diff(t_["stage_ptr"]))
            t_["p_end"] = np.ascontiguousarray(np.minimum(pe, lim.reshape(-1, 16)[wg_of_stage]).reshape(-1).astype(np.int32))
        print(f"{shape}: TIMING ONLY: every block truncated to its first {cap} products", flush=True)
    for merged, inv8 in ((1, 0), (0, 0), (8 | 1, 1), (8, 1)):
        set_tiles(h, tiles2 if merged & 8 else tiles, merged, inv8)
        h.schur(); got = h.array("hsc"); got_bsc = h.array("bsc")
        same = np.array_equal(got.reshape(-1, 36)[offd], ref.reshape(-1, 36)[offd]) and np.array_equal(got_bsc, ref_bsc)
        if merged & 6: same = "n/a (timing variant: " + ("staging only" if merged & 2 else "compute only") + ")"
        t1 = h.time_kernels(20)["linearize_schur"]
        if same is False:
            d = np.abs(got - ref).reshape(-1, 36).max(1) * offd; nbad = int((d > 0).sum())
            wrong = np.nonzero(d > 0)[0]
            pos = {int(b): i for i, b in enumerate(od) if b >= 0}
            print(f"   {nbad} of {len(d)} blocks differ; first wrong blocks {wrong[:8].tolist()} at list positions {[pos.get(int(b), -1) for b in wrong[:8]]}, products {[(int(pbeg[b]), int(pend[b])) for b in wrong[:4]]}", flush=True)
        print(f"{shape}: LDS-staged block pass ({'PIPELINED, ' if merged & 8 else ''}{'one launch with the pose pass' if merged & 1 else 'own launch'}, inverses from {'64-byte rows' if inv8 else 'the landmark systems'}): "
              f"linearise + Schur {t1 * 1e3:7.2f} us, reduced system bit-identical: {same}" + ("" if same is not False else f" (max abs diff {(np.abs(got - ref).reshape(-1, 36)[offd]).max():.3e})"), flush=True)
    set_tiles(h, None, 1, 0)
    h.close()
