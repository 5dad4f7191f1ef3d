"""Randomised checks of the round-6 training entry points against their separate-call forms (development tool):
  yk_gemm_bn_fwd_f32 / yk_dw3x3_bn_fwd_f32 vs gemm / depthwise + yk_bn_train_fwd_res_f32;  yk_bn_train_bwd_f32 (all its paths) vs float64;
  yk_conv3x3_* vs im2col + gemm (+ col2im);  yk_gemm_f32_grouped / yk_dw3x3_bwd_weight_grouped_f32 vs one call per problem.
    python tools/train_fuzz.py [seconds=120] [seed=0]"""
import ctypes as C, os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
import torch
from k210_yolo_framework_amd import engine, netspec as ns
engine.require_gpu()
L = engine.lib()
P = engine._ptr
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = n = 0
def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / max(1e-6, float(b.abs().max())))
def fail(what, info):
    global bad
    bad += 1
    print('MISMATCH', what, info, flush=True)
ACTS = [(ns.ACT_NONE, 0.0), (ns.ACT_RELU, 0.0), (ns.ACT_RELU6, 6.0), (ns.ACT_LEAKY, 0.1)]
t0 = time.time()
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed0 + n)
    n += 1
    kind = rng.integers(5)
    act, alpha = ACTS[rng.integers(4)]
    def bnargs(Cc, M, res):
        g, b = cu(rng.uniform(0.5, 2, Cc)), cu(rng.normal(size=Cc))
        r = cu(rng.normal(size=(M, Cc))) if res else None
        def make():
            y = torch.empty(M, Cc, device='cuda'); sm = torch.empty(Cc, device='cuda'); si = torch.empty(Cc, device='cuda')
            mm = torch.zeros(Cc, device='cuda'); mv = torch.ones(Cc, device='cuda')
            return (y, sm, si, mm, mv), (P(g), P(b), C.c_float(1e-3), int(act), C.c_float(alpha), P(y), P(sm), P(si), P(mm), P(mv), C.c_float(0.99),
                                         P(r) if res else None, st())
        return make, (g, b, r)
    if kind == 0:                                           # conv1x1 + BN forward
        M = int(rng.choice([1, 7, 64, 300, 1120, 1536, 1537, 4480, 9000, 30000])); N = int(rng.choice([1, 3, 4, 8, 24, 75, 96, 130, 320])); K = int(rng.choice([4, 16, 27, 96, 576, 960]))
        X, W = cu(rng.normal(size=(M, K)) + 0.2), cu(rng.normal(size=(N, K)))
        make, keep = bnargs(N, M, rng.random() < 0.5)
        outs = []
        for fused in (0, 1):
            z = torch.empty(M, N, device='cuda'); o, a = make()
            if fused:
                rc = L.yk_gemm_bn_fwd_f32(M, N, K, P(X), K, P(W), K, P(z), *a)
            else:
                rc = L.yk_gemm_f32(0, 1, M, N, K, C.c_float(1), P(X), K, P(W), K, C.c_float(0), P(z), N, st()) or L.yk_bn_train_fwd_res_f32(P(z), C.c_longlong(M), N, *a)
            assert rc == 0, L.yk_last_error()
            outs.append((z,) + o)
        if not torch.equal(outs[0][0], outs[1][0]) or max(rel(u, v) for u, v in zip(outs[1][2:], outs[0][2:])) > 2e-6 or rel(outs[1][1], outs[0][1]) > 2e-5:
            fail('gemm_bn_fwd', (M, N, K, act))
    elif kind == 1:                                         # depthwise + BN forward, backward weight grouped
        B, Hi, Wi = int(rng.integers(1, 5)), int(rng.integers(3, 40)), int(rng.integers(3, 40)); Cc = int(rng.choice([1, 4, 8, 24, 36, 130, 144, 260, 576])); s_ = int(rng.integers(1, 3))
        Ho, Wo = (Hi + 2 - 3) // s_ + 1, (Wi + 2 - 3) // s_ + 1
        M = B * Ho * Wo
        x, w = cu(rng.normal(size=(B, Hi, Wi, Cc))), cu(rng.normal(size=(9, Cc)))
        geom = [C.c_int(v) for v in (B, Hi, Wi, Cc, Ho, Wo, s_, 1, 1)]
        make, keep = bnargs(Cc, M, rng.random() < 0.3)
        outs = []
        for fused in (0, 1):
            z = torch.empty(M, Cc, device='cuda'); o, a = make()
            if fused:
                rc = L.yk_dw3x3_bn_fwd_f32(P(x), P(w), *geom, P(z), *a)
            else:
                rc = L.yk_dw3x3_fwd_f32(P(x), P(w), *geom, P(z), st()) or L.yk_bn_train_fwd_res_f32(P(z), C.c_longlong(M), Cc, *a)
            assert rc == 0, L.yk_last_error()
            outs.append((z,) + o)
        if not torch.equal(outs[0][0], outs[1][0]) or max(rel(u, v) for u, v in zip(outs[1][2:], outs[0][2:])) > 2e-6 or rel(outs[1][1], outs[0][1]) > 2e-5:
            fail('dw_bn_fwd', (B, Hi, Wi, Cc, s_, act))
    elif kind == 2:                                         # BN backward, every path, vs float64
        M = int(rng.choice([2, 70, 1000, 1536, 1537, 4097, 60000])); Cc = int(rng.choice([4, 8, 16, 24, 75, 96, 260]))
        z = (rng.normal(size=(M, Cc)) * rng.uniform(0.5, 3, Cc) + rng.normal(size=Cc)).astype(np.float32)
        g, b, dy = rng.uniform(0.5, 2, Cc).astype(np.float32), rng.normal(size=Cc).astype(np.float32), rng.normal(size=(M, Cc)).astype(np.float32)
        zt = torch.from_numpy(z).double(); mu = zt.mean(0); var = ((zt - mu) ** 2).mean(0); ist = 1 / torch.sqrt(var + 1e-3)
        xh = (zt - mu) * ist; pre = xh * torch.from_numpy(g).double() + torch.from_numpy(b).double()
        if act == ns.ACT_RELU: gate = (pre > 0).double()
        elif act == ns.ACT_RELU6: gate = ((pre > 0) & (pre < 6)).double()
        elif act == ns.ACT_LEAKY: gate = torch.where(pre >= 0, torch.ones_like(pre), torch.full_like(pre, alpha))
        else: gate = torch.ones_like(pre)
        gg = torch.from_numpy(dy).double() * gate
        db, dg = gg.sum(0), (gg * xh).sum(0)
        dz = torch.from_numpy(g).double() * ist * (gg - db / M - xh * dg / M)
        zd, dyd, gd, bd, sm, si = cu(z), cu(dy), cu(g), cu(b), mu.float().cuda(), ist.float().cuda()
        o = torch.empty(M, Cc, device='cuda'); og, ob = torch.empty(Cc, device='cuda'), torch.empty(Cc, device='cuda')
        assert L.yk_bn_train_bwd_f32(P(zd), P(dyd), C.c_longlong(M), Cc, P(gd), P(bd), P(sm), P(si), int(act), C.c_float(alpha), P(o), P(og), P(ob), st()) == 0
        safe = ((pre.abs() > 1e-4) & ((pre - 6).abs() > 1e-4)).all()
        if safe and (rel(o.cpu(), dz) > 3e-4 or rel(og.cpu(), dg) > 3e-4 or rel(ob.cpu(), db) > 3e-4):
            fail('bn_bwd', (M, Cc, act, rel(o.cpu(), dz), rel(og.cpu(), dg), rel(ob.cpu(), db)))
    elif kind == 3:                                         # implicit conv3x3
        B, Hi, Wi = int(rng.integers(1, 4)), int(rng.integers(3, 24)), int(rng.integers(3, 24)); Ci = int(rng.choice([4, 8, 12, 64])); Co = int(rng.choice([4, 8, 20, 75])); s_ = int(rng.integers(1, 3))
        pt, pb = (1, 1) if rng.random() < 0.6 else (1, 0)
        Ho, Wo = (Hi + pt + pb - 3) // s_ + 1, (Wi + pt + pb - 3) // s_ + 1
        if Ho < 1 or Wo < 1:
            continue
        M, KK = B * Ho * Wo, 9 * Ci
        x, w, dy = cu(rng.normal(size=(B, Hi, Wi, Ci))), cu(rng.normal(size=(Co, KK))), cu(rng.normal(size=(M, Co)))
        geom = [C.c_int(v) for v in (B, Hi, Wi, Ci, Ho, Wo, s_, pt, pt)]
        nobn = [None, None, C.c_float(0), 0, C.c_float(0), None, None, None, None, None, C.c_float(0), None]
        z, z2, col = torch.empty(M, Co, device='cuda'), torch.empty(M, Co, device='cuda'), torch.empty(M, KK, device='cuda')
        assert L.yk_conv3x3_bn_fwd_f32(P(x), P(w), *geom, Co, P(z), *nobn, st()) == 0, L.yk_last_error()
        assert L.yk_im2col3x3_f32(P(x), *geom, P(col), st()) == 0
        assert L.yk_gemm_f32(0, 1, M, Co, KK, C.c_float(1), P(col), KK, P(w), KK, C.c_float(0), P(z2), Co, st()) == 0
        if not torch.equal(z, z2):
            fail('conv3x3 fwd', (B, Hi, Wi, Ci, Co, s_, pt, pb))
        if Co % 4 == 0:
            gw, gw2 = torch.empty(Co, KK, device='cuda'), torch.empty(Co, KK, device='cuda')
            assert L.yk_conv3x3_bwd_weight_f32(P(x), P(dy), *geom, Co, P(gw), st()) == 0, L.yk_last_error()
            assert L.yk_gemm_f32(1, 0, Co, KK, M, C.c_float(1), P(dy), Co, P(col), KK, C.c_float(0), P(gw2), KK, st()) == 0
            if not torch.equal(gw, gw2):
                fail('conv3x3 wgrad', (B, Hi, Wi, Ci, Co, s_, pt, pb))
            if s_ == 1:
                dx, dx2 = torch.empty(B, Hi, Wi, Ci, device='cuda'), torch.empty(B, Hi, Wi, Ci, device='cuda')
                assert L.yk_conv3x3_bwd_data_f32(P(dy), P(w), *geom, Co, P(dx), st()) == 0, L.yk_last_error()
                assert L.yk_gemm_f32(0, 0, M, KK, Co, C.c_float(1), P(dy), Co, P(w), KK, C.c_float(0), P(col), KK, st()) == 0
                assert L.yk_col2im3x3_f32(P(col), *geom, P(dx2), st()) == 0
                if rel(dx, dx2) > 2e-5:
                    fail('conv3x3 dgrad', (B, Hi, Wi, Ci, Co, pt, pb, rel(dx, dx2)))
    else:                                                   # grouped weight-gradient GEMMs
        cnt = int(rng.integers(1, 50))
        shp = [(int(rng.choice([4, 16, 24, 75, 96, 320])), int(rng.choice([4, 8, 32, 144, 960])), int(rng.choice([70, 1120, 4480, 17920]))) for _ in range(cnt)]
        As = [cu(rng.normal(size=(K, M))) for (M, N, K) in shp]; Bs = [cu(rng.normal(size=(K, N))) for (M, N, K) in shp]
        Cs = [torch.empty(M, N, device='cuda') for (M, N, K) in shp]; Rs = [torch.empty(M, N, device='cuda') for (M, N, K) in shp]
        for (M, N, K), a, b, r in zip(shp, As, Bs, Rs):
            assert L.yk_gemm_f32(1, 0, M, N, K, C.c_float(1), P(a), M, P(b), N, C.c_float(0), P(r), N, st()) == 0
        ia = lambda v: (C.c_int * cnt)(*v); pa = lambda ts: (C.c_void_p * cnt)(*[t.data_ptr() for t in ts])
        assert L.yk_gemm_f32_grouped(cnt, 1, 0, ia([s[0] for s in shp]), ia([s[1] for s in shp]), ia([s[2] for s in shp]), C.c_float(1), pa(As), ia([s[0] for s in shp]),
                                     pa(Bs), ia([s[1] for s in shp]), C.c_float(0), pa(Cs), ia([s[1] for s in shp]), st()) == 0, L.yk_last_error()
        worst = max(rel(c, r) for c, r in zip(Cs, Rs))
        if worst > 3e-5:
            fail('grouped gemm', (cnt, worst))
torch.cuda.synchronize()
print(f'train_fuzz: {n} random cases, {bad} mismatches, {time.time() - t0:.0f} s')
