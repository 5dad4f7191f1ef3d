import torch, time
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for mb in (7, 27.5, 55, 110, 440):
    n = int(mb * 1e6 / 2)
    bufs = [torch.empty(n, dtype=torch.float16, device='cuda') for _ in range(8)]
    i = [0]
    def cp():
        a = bufs[i[0] % 8]; b = bufs[(i[0] + 1) % 8]; i[0] += 2
        b.copy_(a)
    us = t(cp)
    print(f'copy {mb} MB (rd+wr {2*mb} MB): {us:.1f} us -> {2*mb*1e6/us/1e6:.2f} TB/s')
    def rd():
        a = bufs[i[0] % 8]; i[0] += 1
        return a.sum()
    us = t(rd)
    print(f'read-reduce {mb} MB: {us:.1f} us -> {mb*1e6/us/1e6:.2f} TB/s')
    def wr():
        a = bufs[i[0] % 8]; i[0] += 1
        a.fill_(1.0)
    us = t(wr)
    print(f'fill {mb} MB: {us:.1f} us -> {mb*1e6/us/1e6:.2f} TB/s')
x = torch.empty(1, device='cuda')
print('tiny kernel back-to-back: %.2f us' % t(lambda: x.add_(1), 200))
