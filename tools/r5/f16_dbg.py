"""debug: the fp16 forward, op by op (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "uniir_amd", "src"))
import torch
from uniir_amd import ops
dev = "cuda"
torch.manual_seed(0)
for (M, N, K) in ((512, 768, 512), (64, 512, 256), (100, 512, 640)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    xh, wh = x.half(), w.half()
    ref = xh.float() @ wh.float().t() + b
    y = torch.empty(M, N, device=dev, dtype=torch.float16)
    ops.gemm(xh, wh, y, M, N, K, K, K, N, epilogue=ops.EPI_BF16, bias=b, dtype=ops.DT_F16)
    print("gemm f16 EPI_BF16", (M, N, K), float((y.float() - ref).abs().max()), float(ref.abs().max()))
    y2 = torch.empty(M, N, device=dev, dtype=torch.float16)
    ops.gemm(xh, wh, y2, M, N, K, K, K, N, epilogue=ops.EPI_ACT_ONLY, bias=b, dtype=ops.DT_F16)
    r2 = ref.half().float(); r2 = r2 * torch.sigmoid(1.702 * r2)
    print("gemm f16 ACT_ONLY", float((y2.float() - r2).abs().max()))
    res = torch.randn(M, N, device=dev); y3 = torch.empty(M, N, device=dev)
    ops.gemm(xh, wh, y3, M, N, K, K, K, N, epilogue=ops.EPI_RESID_F32, bias=b, resid=res, dtype=ops.DT_F16)
    print("gemm f16 RESID", float((y3 - ref - res).abs().max()))
from types import SimpleNamespace
from models.uniir_clip.clip_scorefusion.clip_sf import CLIPScoreFusion
os.environ["UNIIR_ALLOW_RANDOM_INIT"] = "1"
model = CLIPScoreFusion("ViT-B/32", device=dev, config=SimpleNamespace(model=SimpleNamespace(gather_embeddings=False), data_config=SimpleNamespace(in_batch_neg_num=0))).float().eval()
img = torch.randn(4, 3, 224, 224, device=dev)
tok = torch.zeros(4, 77, dtype=torch.int32, device=dev); tok[:, 0] = 49406; tok[:, 1:6] = torch.randint(1, 40000, (4, 5), device=dev); tok[:, 6] = 49407
with torch.no_grad():
    for prec in ("bf16", "fp16", "fp32"):
        model.clip_model.precision = prec
        ei = model.encode_image(img); et = model.encode_text(tok)
        print(prec, "img", ei[:, :4].cpu().numpy().round(4).tolist()[0:2], float(ei.norm()), "txt", et[:, :3].cpu().numpy().round(4).tolist()[0], float(et.norm()))
cm = model.clip_model
cm.precision = "fp16"
fl = cm._sync_half_shadow()
d = cm.tower_desc("image", half=True)
print("dtype16", d.dtype16, "stash", d.stash_act, "h16 vs p32", float((fl["h16"].float() - fl["p32"]).abs().max()), float(fl["p32"].abs().max()))
print("conv16h", float(cm._conv16h.float().abs().max()), float(cm._conv16.float().abs().max()))
import ctypes as C
from uniir_amd import _lib
print(C.sizeof(_lib.ClipTower))
lib = _lib.load()
for half in (False, True):
    d = cm.tower_desc("image", half=half)
    M = 4
    need = lib.uniir_clip_tower_workspace_bytes(C.byref(d), M, 0)
    ws = torch.zeros(need, device=dev, dtype=torch.uint8)
    emb = torch.full((M, 512), 7.0, device=dev)
    rc = lib.uniir_clip_tower_fwd(C.byref(d), img.data_ptr(), M, emb.data_ptr(), ws.data_ptr(), need, 0, ops._stream())
    torch.cuda.synchronize()
    W = 768
    rows = ws[256:256 + M * W * 4].view(torch.float32).view(M, W)
    pooled = ws[256 + M * W * 4:256 + M * W * 4 + M * W * 2].view(torch.float16 if half else torch.bfloat16).view(M, W)
    print("half", half, "rc", rc, "rows", float(rows.norm()), "pooled", float(pooled.float().norm()), "emb", float(emb.norm()), emb[0, :3].tolist())
    g, b = cm.visual.ln_post.weight.data, cm.visual.ln_post.bias.data
    ref = torch.nn.functional.layer_norm(rows, (W,), g, b, 1e-5)
    raw = ws[256 + M * W * 4:256 + M * W * 4 + M * W * 2]
    print("   pooled vs LN(rows): as f16", float((raw.view(torch.float16).view(M, W).float() - ref).abs().max()),
          "as bf16", float((raw.view(torch.bfloat16).view(M, W).float() - ref).abs().max()))
    proj = cm.visual.proj.data
    print("   emb vs pooled@proj: ", float((emb - ref @ proj).abs().max()), float((ref @ proj).abs().max()))
