"""The slice of the `clip` package surface that UniIR uses (clip.load, clip.tokenize), backed by uniir_amd.

Reference call sites: src/models/uniir_clip/clip_scorefusion/clip_sf.py:25-26 (clip.load -> (model, preprocess);
clip.tokenize).  Upstream downloads weights / ships a BPE vocabulary; neither exists offline, so:
  * load(): builds the architecture for `name` (random init, upstream std choices) and, when
    <download_root>/<name>.pt (a state dict, or a checkpoint with a "state_dict"/"model" entry) exists, loads it;
  * tokenize(): needs <UNIIR_BPE_PATH or download_root>/bpe_simple_vocab_16e6.txt.gz; without the file it raises
    (tokenizer parity is unpinned offline, SURVEY.md section 7 'No network').
"""
import os

import numpy as np
import torch

from .clip_model import CLIP, CLIP_CONFIGS

_MEAN = (0.48145466, 0.4578275, 0.40821073)
_STD = (0.26862954, 0.26130258, 0.27577711)


def _preprocess(n_px):
    def fn(image):
        """PIL image -> float tensor [3, n_px, n_px]: bicubic resize of the short side, centre crop, RGB,
        CLIP mean/std normalisation (upstream clip._transform)."""
        from PIL import Image
        w, h = image.size
        nh, nw, top, left = resize_geometry(h, w, n_px)          # torchvision Resize + CenterCrop integers
        image = image.resize((nw, nh), Image.BICUBIC)
        image = image.crop((left, top, left + n_px, top + n_px)).convert("RGB")
        a = torch.from_numpy(np.asarray(image, dtype=np.float32) / 255.0).permute(2, 0, 1)
        return (a - torch.tensor(_MEAN).view(3, 1, 1)) / torch.tensor(_STD).view(3, 1, 1)

    return fn


def resize_geometry(h, w, n_px, center_crop=True):
    """integer geometry of the transform: torchvision Resize(n_px) (short side n_px, long side int(n_px * long / short)) +
    CenterCrop (offsets round-half-even((size - n_px) / 2)) as upstream clip._transform composes them; center_crop=False
    is BLIP's square Resize((n_px, n_px)) (blip_transform.py:41-48).  -> (oh, ow, top, left)"""
    if not center_crop:
        return n_px, n_px, 0, 0
    if w <= h:
        ow, oh = n_px, int(n_px * h / w)
    else:
        oh, ow = n_px, int(n_px * w / h)
    return oh, ow, int(round((oh - n_px) / 2.0)), int(round((ow - n_px) / 2.0))


def preprocess_on_device(images, n_px, device, mean=_MEAN, std=_STD, center_crop=True, out=None):
    """Decoded RGB images (PIL images or uint8 [h, w, 3] arrays / tensors, any sizes) -> fp32 [M, 3, n_px, n_px] on the
    device through libuniir_hip's uniir_image_preprocess: the resize / crop / normalise chain of the reference's CPU
    workers, bit-exact with Pillow's BICUBIC for the integer stage (include/uniir_hip.h [IMAGE])."""
    import ctypes

    from . import _lib, ops
    lib = _lib.load()
    M = len(images)
    if out is None:
        out = torch.empty(M, 3, n_px, n_px, device=device, dtype=torch.float32)
    mean_c, std_c = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
    stream = torch.cuda.current_stream(device).cuda_stream
    ws = None
    for i, img in enumerate(images):
        if not isinstance(img, (torch.Tensor, np.ndarray)):
            img = np.asarray(img.convert("RGB"), dtype=np.uint8)
        t = torch.as_tensor(img)
        if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
            raise ValueError("images must be RGB uint8 [h, w, 3]")
        t = t.contiguous().to(device, non_blocking=True)
        h, w = int(t.shape[0]), int(t.shape[1])
        oh, ow, top, left = resize_geometry(h, w, n_px, center_crop)
        need = lib.uniir_image_workspace_bytes(h, w, oh, ow, n_px)
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, device=device, dtype=torch.uint8)
        ops.check(lib.uniir_image_preprocess(t.data_ptr(), h, w, oh, ow, top, left, n_px, mean_c, std_c, out[i].data_ptr(),
                                             ws.data_ptr(), ws.numel(), stream), "image_preprocess")
    return out


def load(name="ViT-B/32", device="cuda", jit=False, download_root=None, seed=0):
    if name not in CLIP_CONFIGS:
        raise RuntimeError(f"Model {name} not found; available models = {list(CLIP_CONFIGS)}")
    model = CLIP(CLIP_CONFIGS[name], seed=seed)
    if download_root:
        path = os.path.join(os.path.expanduser(download_root), name.replace("/", "-") + ".pt")
        if os.path.exists(path):
            sd = torch.load(path, map_location="cpu")
            sd = sd.get("state_dict", sd.get("model", sd)) if isinstance(sd, dict) else sd.state_dict()
            sd = {k: v.float() for k, v in sd.items() if k in model.state_dict()}
            model.load_state_dict(sd, strict=True)
    if device is not None and str(device) != "cpu":
        model = model.to(device)
    return model, _preprocess(CLIP_CONFIGS[name]["image_resolution"])


def tokenize(texts, context_length=77, truncate=False):
    raise RuntimeError(
        "clip.tokenize needs OpenAI's bpe_simple_vocab_16e6.txt.gz, which is not available offline; feed token ids "
        "(int32 [n, 77], SOT=49406 ... EOT=49407, zero padded) or install the vocabulary (see INTEGRATION.md)")
