"""The slice of the `clip` package surface that UniIR uses (clip.load, clip.tokenize), backed by uniir_amd.

Reference call sites: src/models/uniir_clip/clip_scorefusion/clip_sf.py:25-26 (clip.load -> (model, preprocess);
clip.tokenize).  Upstream downloads weights / ships a BPE vocabulary; neither exists offline, so:
  * load(): builds the architecture for `name` (random init, upstream std choices) and, when
    <download_root>/<name>.pt (a state dict, or a checkpoint with a "state_dict"/"model" entry) exists, loads it;
  * tokenize(): the byte-level BPE of clip/simple_tokenizer.py (BPETokenizer below, pinned on a synthetic merge list
    against an independent implementation: tests/golden/g15_bpe.json); it needs OpenAI's bpe_simple_vocab_16e6.txt.gz
    at $UNIIR_BPE_PATH, $UNIIR_CLIP_ROOT or ~/.cache/clip and raises without it (parity on the REAL vocabulary is
    unpinned offline, SURVEY.md section 7 'No network').
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


class RawImage:
    """a decoded RGB image waiting for the device transform (uint8 [h, w, 3], shareable between loader workers)"""
    __slots__ = ("data",)

    def __init__(self, data):
        self.data = data


class RawImageTransform:
    """Drop-in `img_preprocess_fn` that defers resize / crop / normalise to the GPU: the dataset worker only decodes, the
    collator builds a RawImageBatch, and the batch is transformed by `preprocess_on_device` when it is moved to the device
    (host_utils.DevicePrefetcher / batch_to_device).  Bit-identical tensors to the CPU transform `_preprocess(n_px)`."""

    def __init__(self, n_px, mean=_MEAN, std=_STD, center_crop=True):
        self.n_px, self.mean, self.std, self.center_crop = n_px, tuple(mean), tuple(std), center_crop

    def __call__(self, image):
        return RawImage(torch.from_numpy(np.array(image.convert("RGB"), dtype=np.uint8)))


class RawImageBatch:
    """what a collator stacks instead of a float tensor when its items are RawImages (None = the black padding image of
    items without an image, which the reference builds as zeros AFTER normalisation)"""

    def __init__(self, images, transform):
        self.images, self.transform = list(images), transform

    def size(self, dim=0):
        n = self.transform.n_px
        return (len(self.images), 3, n, n)[dim]

    def pin_memory(self):
        self.images = [im if im is None else im.pin_memory() for im in self.images]
        return self

    def to_device(self, device):
        t = self.transform
        out = torch.zeros(len(self.images), 3, t.n_px, t.n_px, device=device, dtype=torch.float32)
        have = [i for i, im in enumerate(self.images) if im is not None]
        if have:
            got = preprocess_on_device([self.images[i] for i in have], t.n_px, device, t.mean, t.std, t.center_crop)
            out[torch.tensor(have, device=device)] = got
        return out


def load(name="ViT-B/32", device="cuda", jit=False, download_root=None, seed=0):
    if name not in CLIP_CONFIGS:
        raise RuntimeError(f"Model {name} not found; available models = {list(CLIP_CONFIGS)}")
    model = CLIP(CLIP_CONFIGS[name], seed=seed)
    if download_root:
        path = os.path.join(os.path.expanduser(download_root), name.replace("/", "-") + ".pt")
        if os.path.exists(path):
            try:            # the published files are TorchScript archives (upstream clip.load falls back the other way)
                sd = torch.jit.load(path, map_location="cpu").state_dict()
            except RuntimeError:
                sd = torch.load(path, map_location="cpu", weights_only=False)
            sd = sd.get("state_dict", sd.get("model", sd)) if isinstance(sd, dict) else sd.state_dict()
            sd = {k: v.float() for k, v in sd.items() if k in model.state_dict()}
            model.load_state_dict(sd, strict=True)
        elif os.environ.get("UNIIR_ALLOW_RANDOM_INIT") != "1":
            # upstream clip.load downloads the file or fails; silently fine-tuning from random weights would produce
            # plausible-looking checkpoints and embeddings
            raise FileNotFoundError(f"pretrained CLIP weights not found: {path} (there is no network to download them from; "
                                    "put the openai/CLIP .pt file there, or set UNIIR_ALLOW_RANDOM_INIT=1 for a dry run)")
        else:
            import warnings
            warnings.warn(f"{path} is missing: CLIP {name} stays RANDOMLY INITIALISED (UNIIR_ALLOW_RANDOM_INIT=1)")
    if device is not None and str(device) != "cpu":
        model = model.to(device)
    return model, _preprocess(CLIP_CONFIGS[name]["image_resolution"])


class BPETokenizer:
    """Byte-level BPE of openai/CLIP (clip/simple_tokenizer.py, third-party: restated from its published behaviour, pinned
    in tests against transformers' CLIPTokenizer on a synthetic vocabulary -- the real 16e6 vocabulary is not available
    offline).  Vocabulary order: the 256 byte symbols, the same with the end-of-word marker, one entry per merge rule of
    the file (lines 1 .. 49152 - 256 - 2), then <|startoftext|>, <|endoftext|>."""
    EOW = "</w>"

    def __init__(self, bpe_path):
        import gzip
        import regex
        opener = gzip.open if bpe_path.endswith(".gz") else open
        with opener(bpe_path, "rt", encoding="utf-8") as f:
            lines = f.read().split("\n")
        rules = [tuple(ln.split()) for ln in lines[1:49152 - 256 - 2 + 1] if ln.strip()]
        # printable bytes stand for themselves, the other 68 get code points from 256 upwards (no whitespace / control symbols)
        keep = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
        self.byte_sym, extra = {}, 0
        for b in range(256):
            if b in keep:
                self.byte_sym[b] = chr(b)
            else:
                self.byte_sym[b] = chr(256 + extra)
                extra += 1
        base = [self.byte_sym[b] for b in keep] + [self.byte_sym[b] for b in range(256) if b not in keep]
        vocab = base + [v + self.EOW for v in base] + ["".join(r) for r in rules] + ["<|startoftext|>", "<|endoftext|>"]
        self.encoder = {tok: i for i, tok in enumerate(vocab)}
        self.rank = {r: i for i, r in enumerate(rules)}
        self.cache = {"<|startoftext|>": ("<|startoftext|>",), "<|endoftext|>": ("<|endoftext|>",)}
        self.splitter = regex.compile(
            r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+", regex.IGNORECASE)
        self.sot, self.eot = self.encoder["<|startoftext|>"], self.encoder["<|endoftext|>"]

    def _merge_word(self, token):
        if token in self.cache:
            return self.cache[token]
        parts = list(token[:-1]) + [token[-1] + self.EOW]
        while len(parts) > 1:
            best = min(((self.rank.get((a, b), float("inf")), i) for i, (a, b) in enumerate(zip(parts, parts[1:]))))
            if best[0] == float("inf"):
                break
            first, second = parts[best[1]], parts[best[1] + 1]
            merged, i = [], 0
            while i < len(parts):          # every occurrence of the best-ranked pair, left to right
                if i + 1 < len(parts) and parts[i] == first and parts[i + 1] == second:
                    merged.append(first + second)
                    i += 2
                else:
                    merged.append(parts[i])
                    i += 1
            parts = merged
        self.cache[token] = tuple(parts)
        return self.cache[token]

    @staticmethod
    def clean(text):
        import html
        import re
        try:
            import ftfy
            text = ftfy.fix_text(text)
        except ImportError:          # upstream repairs mojibake first; without ftfy well-formed text passes unchanged
            pass
        text = html.unescape(html.unescape(text)).strip()
        return re.sub(r"\s+", " ", text).strip().lower()

    def encode(self, text):
        ids = []
        for tok in self.splitter.findall(self.clean(text)):
            sym = "".join(self.byte_sym[b] for b in tok.encode("utf-8"))
            ids.extend(self.encoder[p] for p in self._merge_word(sym))
        return ids


_TOKENIZER = None


def _bpe_path():
    cands = [os.environ.get("UNIIR_BPE_PATH", "")]
    for root in (os.environ.get("UNIIR_CLIP_ROOT", ""), os.path.expanduser("~/.cache/clip")):
        cands.append(os.path.join(root, "bpe_simple_vocab_16e6.txt.gz"))
    for c in cands:
        if c and os.path.isdir(c):
            c = os.path.join(c, "bpe_simple_vocab_16e6.txt.gz")
        if c and os.path.isfile(c):
            return c
    return None


def tokenize(texts, context_length=77, truncate=False):
    """clip.tokenize: [SOT] + BPE ids + [EOT], zero padded to context_length -> int32 [n, context_length]; too long inputs
    raise unless truncate (then cut and the last id forced to EOT)."""
    global _TOKENIZER
    if _TOKENIZER is None:
        path = _bpe_path()
        if path is None:
            raise RuntimeError(
                "clip.tokenize needs OpenAI's bpe_simple_vocab_16e6.txt.gz, which is not available offline; set "
                "UNIIR_BPE_PATH to the file (or feed token ids: int32 [n, 77], SOT=49406 ... EOT=49407, zero padded)")
        _TOKENIZER = BPETokenizer(path)
    tk = _TOKENIZER
    if isinstance(texts, str):
        texts = [texts]
    out = torch.zeros(len(texts), context_length, dtype=torch.int32)
    for i, text in enumerate(texts):
        ids = [tk.sot] + tk.encode(text) + [tk.eot]
        if len(ids) > context_length:
            if not truncate:
                raise RuntimeError(f"Input {text} is too long for context length {context_length}")
            ids = ids[:context_length]
            ids[-1] = tk.eot
        out[i, :len(ids)] = torch.tensor(ids, dtype=torch.int32)
    return out
