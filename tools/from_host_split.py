"""Where the from-host step loses against the resident one: the same engine.Pipeline (4 batches in flight, replayed graphs) timed as
  resident        frames in HBM, detections in HBM                           (bench.py `value`)
  h2d_only        frames from pinned host memory (copy stream), detections stay in HBM
  d2h_only        frames in HBM, detections written by the compaction kernel into pinned host memory
  from_host       both legs                                                  (bench.py `value_from_host`)
python tools/from_host_split.py [steps=300]"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch
from k210_yolo_framework_amd import engine, netspec
from k210_yolo_framework_amd.helper import VOC_ANCHORS

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = 32
spec = netspec.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75)
w = spec.init_weights(seed=1)
frames = torch.randint(0, 256, (B, 224, 320, 3), dtype=torch.uint8, device='cuda', generator=torch.Generator(device='cuda').manual_seed(0))


def timed(fn, pipe):
    for _ in range(16):
        fn()
    torch.cuda.synchronize()
    best = 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        best = max(best, B * steps / (time.perf_counter() - t0))
    return best


res = {}
for mode in (sys.argv[2].split(',') if len(sys.argv) > 2 else ('resident', 'h2d_only', 'h2d_hostwait', 'h2d_hostwait_prefetch', 'from_host', 'from_host_hostwait', 'resident')):
    pipe = engine.Pipeline(spec, w, VOC_ANCHORS, max_batch=B, depth=4, precision='f16x2', graph=True)
    for i in range(4):
        pipe.host_input(i).copy_(frames.cpu())
    if mode == 'resident':
        fn = lambda: pipe.submit(frames, sync_input=False)
    elif mode == 'from_host':
        fn = lambda: pipe.submit_host(None)
    elif mode in ('h2d_copy_nowait', 'h2d_wait_nocopy'):
        import ctypes as C
        L = engine.lib()

        def fn(mode=mode):
            s = pipe.slots[pipe._n % pipe.depth]
            pipe._n += 1
            pipe._host_side(s)
            b = s.h2d_next
            s.h2d_next ^= 1
            cs = pipe._copy_stream
            if mode == 'h2d_copy_nowait':                              # the copy runs beside the compute streams, nobody waits for it
                L.yk_memcpy_async(C.c_void_p(s.h2d_bufs[b].data_ptr()), C.c_void_p(s.h_src.data_ptr()), C.c_size_t(B * s.src[0].numel()), C.c_void_p(cs.cuda_stream))
            else:                                                      # no copy: only the cross-queue dependency (event on the copy stream)
                s.h2d_copied[b].record(cs)
                s.stream.wait_event(s.h2d_copied[b])
            pipe._run(s, B, s.src.data_ptr(), False, False, 0.7, 0.5, 30, False)
    elif mode in ('h2d_hostwait', 'h2d_hostwait_prefetch', 'from_host_hostwait'):
        import ctypes as C
        L = engine.lib()
        state = {'pending': None}

        def copy(s):
            pipe._host_side(s)
            b = s.h2d_next
            s.h2d_next ^= 1
            if s.h2d_free[b] is not None and not s.h2d_free[b].query():
                s.h2d_free[b].synchronize()
            L.yk_memcpy_async(C.c_void_p(s.h2d_bufs[b].data_ptr()), C.c_void_p(s.h_src.data_ptr()), C.c_size_t(B * s.src[0].numel()), C.c_void_p(pipe._copy_stream.cuda_stream))
            s.h2d_copied[b].record(pipe._copy_stream)
            s.h2d_last = b
            return b

        def fn(mode=mode):
            s = pipe.slots[pipe._n % pipe.depth]
            pipe._n += 1
            if mode == 'h2d_hostwait_prefetch':
                # the copy of THIS batch was issued during the previous submit; issue the next slot's now, then wait for ours on the host
                if state['pending'] is None:
                    state['pending'] = (s, copy(s))
                (s0, b) = state['pending']
                nxt = pipe.slots[pipe._n % pipe.depth]
                state['pending'] = (nxt, copy(nxt))
                s0.h2d_copied[b].synchronize()
                pipe._run(s0, B, s0.h2d_bufs[b].data_ptr(), False, False, 0.7, 0.5, 30, False)
                ev = torch.cuda.Event(); ev.record(s0.stream); s0.h2d_free[b] = ev
                return
            b = copy(s)
            s.h2d_copied[b].synchronize()                              # the HOST waits for the copy: no barrier packet in the compute queue
            if mode == 'from_host_hostwait':
                pipe._h2d = lambda s_, Bq, p=s.h2d_bufs[b].data_ptr(): p
                pipe._h2d_done = lambda s_: None
                pipe._run(s, B, None, True, False, 0.7, 0.5, 30, False)
            else:
                pipe._run(s, B, s.h2d_bufs[b].data_ptr(), False, False, 0.7, 0.5, 30, False)
            ev = torch.cuda.Event(); ev.record(s.stream); s.h2d_free[b] = ev
    elif mode.startswith('copy_'):
        import ctypes as C
        L = engine.lib()
        pipe._host_side(pipe.slots[0])
        scratch = torch.empty((2, B, 224, 320, 3), dtype=torch.uint8, device='cuda')
        src_d = torch.randint(0, 255, (B, 224, 320, 3), dtype=torch.uint8, device='cuda')
        nb = B * 224 * 320 * 3
        kinds = {'copy_h2d': (pipe.slots[0].h_src.data_ptr(), nb, 1), 'copy_half': (pipe.slots[0].h_src.data_ptr(), nb // 2, 1),
                 'copy_2x': (pipe.slots[0].h_src.data_ptr(), nb, 2), 'copy_d2d': (src_d.data_ptr(), nb, 1), 'copy_tiny': (pipe.slots[0].h_src.data_ptr(), 4096, 1)}
        for nch in (2, 3, 4, 8, 16):
            kinds[f'copy_chunks{nch}'] = (pipe.slots[0].h_src.data_ptr(), nb, -nch)
        srcp, nbytes, reps = kinds[mode]
        cnt = [0]

        def fn():
            if reps < 0:                                               # the same bytes as -reps back-to-back copies
                n = -reps
                per = (nbytes + n - 1) // n
                for r in range(n):
                    o = r * per
                    L.yk_memcpy_async(C.c_void_p(scratch[cnt[0] & 1].data_ptr() + o), C.c_void_p(srcp + o), C.c_size_t(min(per, nbytes - o)), C.c_void_p(pipe._copy_stream.cuda_stream))
            else:
                for r in range(reps):                                  # traffic only: nobody reads the destination, nobody waits
                    L.yk_memcpy_async(C.c_void_p(scratch[(cnt[0] + r) & 1].data_ptr()), C.c_void_p(srcp), C.c_size_t(nbytes), C.c_void_p(pipe._copy_stream.cuda_stream))
            cnt[0] += 1
            pipe.submit(frames, sync_input=False)
    elif mode == 'h2d_only':
        def fn():
            s = pipe.slots[pipe._n % pipe.depth]
            pipe._n += 1
            ptr = pipe._h2d(s, B)
            pipe._run(s, B, ptr, False, False, 0.7, 0.5, 30, False)
            pipe._h2d_done(s)
    else:
        pipe._h2d = lambda s, Bq: s.src.data_ptr()
        pipe._h2d_done = lambda s: None
        for sl in pipe.slots:
            sl.src.copy_(frames)
        fn = lambda: pipe.submit_host(None)
    r = timed(fn, pipe)
    res.setdefault(mode, []).append(round(r))
    print(mode, round(r), flush=True)
    pipe.close()
print(res)
