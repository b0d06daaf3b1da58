import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image2text_amd import ops
BF16 = torch.bfloat16
dev = torch.device('cuda:0')
M, N, K = 131072, 1536, 768
a = (torch.randn(M, K, device=dev) * 0.5).to(BF16)
w = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
bias = torch.randn(N, device=dev)
outs = {}
for mode in ('0', '2'):
    os.environ['I2T_GEMM3'] = mode
    c = torch.zeros(M, N, dtype=BF16, device=dev)
    ops.gemm(a, w, c, M, N, K, bias=bias)
    torch.cuda.synchronize()
    outs[mode] = c.float()
d = (outs['2'] - outs['0']).abs() > 0.1
print('mismatch count', int(d.sum()), 'of', d.numel())
rows = d.any(1).nonzero().flatten()
cols = d.any(0).nonzero().flatten()
print('rows with mismatch', rows.numel(), 'cols', cols.numel())
print('row % 256 histogram (16-row groups):', torch.bincount((rows % 256) // 16, minlength=16).tolist())
print('col % 128 histogram (16-col groups):', torch.bincount((cols % 128) // 16, minlength=8).tolist())
tm = torch.bincount(rows // 256, minlength=M // 256)
print('row tiles affected', int((tm > 0).sum()), 'of', M // 256, 'first few', (tm > 0).nonzero().flatten()[:20].tolist())
r0 = int(rows[0]); cs = d[r0].nonzero().flatten()
print('row', r0, 'bad cols', cs[:16].tolist(), 'n', cs.numel())
c0 = int(cs[0])
print('got', outs['2'][r0, c0].item(), 'want', outs['0'][r0, c0].item(), 'bias', bias[c0].item(), 'diff', (outs['2'][r0, c0] - outs['0'][r0, c0]).item())
# is the bad value the no-bias value, or a value from elsewhere?
nb = (a[r0].float() @ w[c0].float()).item()
print('no-bias value', nb)
