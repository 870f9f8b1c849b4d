"""Drop-in for UniIR src/models/uniir_clip/clip_featurefusion/clip_ff.py (class CLIPFeatureFusion): same constructor,
methods, batch / output dictionaries and state-dict keys (`clip_model.*` without `text_projection`, `t5_layers.block...`);
the towers without pooling, the T5 fusion stack, the mean pooling and the loss run on libuniir_hip.so (MI355X).

Reference lines mirrored: __init__ :63-110 (ViT-B/32 -> T5 d_model 512, ViT-L/14 -> 768; 2 layers, 12 heads, d_kv 64,
T5Config defaults d_ff 2048 / ReLU / 32 buckets), encode_text :148-156, encode_image :158-159, encode_multimodal_input
:161-192 (the masks are accepted and unused, like there), compute_inbatch_contrastive_loss :194-265 (identical to
CLIP_SF's, incl. the hard-negative branch), encode_mbeir_batch :272-298.  T5 dropout 0.1 is not applied.
"""
import torch
from torch import nn

from uniir_amd import clip_front
from uniir_amd.blip_model import _attach
from uniir_amd.clipff_model import FusionFn, t5_param_shapes
from uniir_amd.losses import HardNegNCEFn, InBatchNCEFn

_T5_DMODEL = {"ViT-B/32": 512, "ViT-L/14": 768}


class _T5Group:
    """the T5 stack as one optimizer group over its own flat store (clip_featurefusion/train.py:52-61)"""

    def __init__(self, owner, lr, weight_decay):
        self.owner, self.lr, self.weight_decay = owner, lr, weight_decay
        self.params = [p for _, p in owner._t5_named()]

    def store(self):
        return self.owner._ensure_t5()


class CLIPFeatureFusion(nn.Module):
    def __init__(self, model_name="ViT-B/32", device="cuda", jit=False, download_root=None, config=None, t5_config=None):
        super().__init__()
        self.clip_model, self.img_preprocess_fn = clip_front.load(model_name, device, jit, download_root=download_root)
        self.tokenizer = clip_front.tokenize
        self.loss_function = nn.CrossEntropyLoss()  # attribute parity; the fused kernel computes the CE
        t5 = dict(d_model=_T5_DMODEL.get(model_name), num_heads=12, d_ff=2048, num_layers=2, dropout_rate=0.1)
        t5.update(t5_config or {})
        self.t5_dropout = float(t5["dropout_rate"])     # T5Config() default; active in train mode only
        if t5["d_model"] is None:
            raise NotImplementedError("Only ViT-B/32 and ViT-L/14 are supported.")
        self.t5_heads, self.t5_layers_n, self._t5_cfg = t5["num_heads"], t5["num_layers"], t5
        self._t5_shapes = t5_param_shapes(t5["d_model"], t5["num_heads"], 64, t5["d_ff"], t5["num_layers"])
        g = torch.Generator().manual_seed(1)
        self.t5_layers = nn.Module()
        for n, shp in self._t5_shapes:
            v = torch.ones(shp) if n.endswith("layer_norm.weight") else torch.randn(shp, generator=g) * shp[-1] ** -0.5
            _attach(self.t5_layers, n, nn.Parameter(v))
        if config is not None:
            self.gather_embeddings = config.model.gather_embeddings
            self.in_batch_neg_num = config.data_config.in_batch_neg_num
        else:
            self.gather_embeddings = None
            self.in_batch_neg_num = None
        del self.clip_model.text_projection      # clip_ff.py:104 (unused: the text tower is not pooled / projected)
        self._t5 = None
        self._t5_version = -1
        if device is not None and str(device) != "cpu":
            self.to(device)

    # ---- T5 flat store ---------------------------------------------------------------------------------------
    def _t5_named(self):
        return [(n, self.t5_layers.get_parameter(n)) for n, _ in self._t5_shapes]

    def _ensure_t5(self):
        from uniir_amd.blip_model import FlatStore
        dev = self.clip_model.logit_scale.device
        if dev.type != "cuda":
            raise RuntimeError("uniir_amd CLIPFeatureFusion runs on an MI355X only (no CPU path); move the model to cuda")
        st = self._t5
        named = self._t5_named()
        if st is None or st.p32.device != dev or any(p.data_ptr() != st.p32.data_ptr() + 4 * st.off[n] for n, p in named):
            st = FlatStore(self._t5_shapes, dev, True)
            for n, p in named:
                st.p(n).copy_(p.data.float())
                p.data = st.p(n)
                p.grad = st.grad_view(n)
            self._t5 = st
            self._t5_version = -1
        ver = sum(p._version for _, p in named)
        if ver != self._t5_version:
            st.refresh_shadow()
            self._t5_version = ver
        return st

    def t5_optimizer_group(self, lr, weight_decay=0.2):
        return _T5Group(self, lr, weight_decay)

    def zero_grad(self, set_to_none=False):
        self.clip_model.zero_grad()
        if self._t5 is not None:
            self._t5.g32.zero_()
            for n, p in self._t5_named():
                p.grad = self._t5.grad_view(n)

    # ---- reference surface -----------------------------------------------------------------------------------
    def get_img_preprocess_fn(self):
        return self.img_preprocess_fn

    def get_tokenizer(self):
        def tokenizer_wrapper(txt):
            return self.tokenizer(txt, context_length=77, truncate=True)

        return tokenizer_wrapper

    def encode_multimodal_input(self, txt_tensor, img_tensor, txt_mask=None, img_mask=None):
        self.clip_model._sync_shadow()
        self._ensure_t5()
        anchor = torch.zeros(1, device=img_tensor.device, requires_grad=torch.is_grad_enabled())
        return FusionFn.apply(self, txt_tensor.to(torch.int32).contiguous(), img_tensor.contiguous(), anchor)

    def get_logit_scale(self):
        return self.clip_model.logit_scale.exp()

    def compute_inbatch_contrastive_loss(self, batch):
        index_mapping = batch["index_mapping"]
        embeddings = self.encode_multimodal_input(batch["txt_batched"], batch["image_batched"],
                                                  batch["txt_mask_batched"], batch["image_mask_batched"])
        dev = embeddings.device
        idx_q = torch.tensor(index_mapping["query"], dtype=torch.int32).flatten().to(dev, non_blocking=True)
        idx_p = torch.tensor(index_mapping["pos_cand"], dtype=torch.int32).flatten().to(dev, non_blocking=True)
        if "neg_cand_list" in index_mapping:
            idx_n = torch.tensor(index_mapping["neg_cand_list"], dtype=torch.int32).flatten().to(dev, non_blocking=True)
            loss, accuracy = HardNegNCEFn.apply(embeddings, idx_q, idx_p, idx_n, self.get_logit_scale(),
                                                int(self.in_batch_neg_num or 0))
            return {"loss": loss, "accuracy": accuracy}
        loss, accuracy, _ = InBatchNCEFn.apply(embeddings, idx_q, idx_p, self.get_logit_scale(), bool(self.gather_embeddings))
        return {"loss": loss, "accuracy": accuracy}

    def forward(self, batch, encode_mbeir_batch=False):
        if encode_mbeir_batch:
            return self.encode_mbeir_batch(batch)
        return self.compute_inbatch_contrastive_loss(batch)

    def encode_mbeir_batch(self, batch):
        id_list = batch.get("did_list") or batch.get("qid_list")
        assert id_list is not None, "id_list must be provided."
        assert isinstance(id_list[0], int), "id_list must be hashed to int."
        embeddings = self.encode_multimodal_input(batch["txt_batched"], batch["image_batched"],
                                                  batch["txt_mask_batched"], batch["image_mask_batched"])
        assert embeddings.size(0) == len(id_list), "embeddings and id_batched must have the same batch size."
        return embeddings, id_list
