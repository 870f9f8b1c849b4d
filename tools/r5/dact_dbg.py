import torch, sys
sys.path.insert(0, "/root/repo")
from uniir_amd import ops
DEV="cuda"
bf=lambda t: t.bfloat16()
torch.manual_seed(31)
m,n,k,act=1024,512,256,0
x, w = bf(torch.randn(m, k, device=DEV)), bf(torch.randn(n, k, device=DEV) * 0.2)
b = torch.randn(n, device=DEV)
g_fwd = torch.empty(m, n, device=DEV, dtype=torch.bfloat16)
f = ops.linear_fwd(x, w, b, epilogue=ops.EPI_BIAS_ACT, C2=g_fwd, act=act)
n2 = 256
dy, w2 = bf(torch.randn(m, n2, device=DEV)), bf(torch.randn(n2, n, device=DEV) * 0.2)
cs1, cs2 = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
g_bwd = torch.empty(m, n, device=DEV, dtype=torch.bfloat16)
dx1 = ops.linear_dgrad(dy, w2, aux=f, act=act, act_out=g_bwd, colsum=cs1)
dx2 = ops.linear_dgrad(dy, w2, aux=f, act=act, colsum=cs2)
d = (dx1.float()-dx2.float())
print("dx equal", torch.equal(dx1, dx2), "ndiff", int((d!=0).sum()), "max", float(d.abs().max()))
print("cs rel", float((cs1-cs2).abs().max()/cs2.abs().max()), float((cs1-cs2).norm()/cs2.norm()))
diff = (g_bwd.float() - g_fwd.float()).abs()
print("g diff frac", float((diff > 0).float().mean()))
idx = (d!=0).nonzero()[:5]
print(idx)
a = f.float()
sg = torch.sigmoid(1.702 * a)
exact = (dy.float() @ w2.float()) * (sg * (1 + 1.702 * a * (1 - sg)))
for name, dx in (("dx1(C2)", dx1), ("dx2", dx2)):
    e = (dx.float() - exact)
    bad = ~torch.isfinite(dx.float()) | (e.abs() > 0.05 * exact.abs() + 0.05)
    print(name, "nonfinite", int((~torch.isfinite(dx.float())).sum()), "bad", int(bad.sum()))
    ii = bad.nonzero()[:8]
    for r, c in ii.tolist():
        print("   ", r, c, float(dx[r, c]), float(exact[r, c]), "f=", float(f[r, c]))
gexact = a * sg
bad = (g_bwd.float() - gexact).abs() > 0.02 * gexact.abs() + 0.02
print("g_bwd bad", int(bad.sum()), bad.nonzero()[:6].tolist())
print("rows with bad dx1:", sorted(set((~torch.isfinite(dx1.float())).nonzero()[:, 0].tolist()))[:40])
