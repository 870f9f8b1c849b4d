"""Drop-in for UniIR src/models/uniir_clip/clip_scorefusion/clip_sf.py (class CLIPScoreFusion): same constructor,
methods, batch dictionary and output dictionary; the towers and the loss run on libuniir_hip.so (MI355X).

Reference lines mirrored: __init__ :14-30, get_tokenizer :35-41, encode_multimodal_input :53-63,
get_logit_scale :65-66, compute_inbatch_contrastive_loss :68-147, forward :149-152, encode_mbeir_batch :154-168.
The hard-negative branch (:105-131) runs on `uniir_hardneg_{fwd,bwd}` (one workgroup per query, fp32).
"""
import torch
from torch import nn

from uniir_amd import clip_front
from uniir_amd.clip_model import _tensor_version
from uniir_amd.losses import FuseFn, HardNegNCEFn, InBatchNCEFn


class CLIPScoreFusion(nn.Module):
    def __init__(self, model_name="ViT-B/32", device="cuda", jit=False, download_root=None, config=None):
        super().__init__()
        self.clip_model, self.img_preprocess_fn = clip_front.load(model_name, device, jit, download_root=download_root)
        self.tokenizer = clip_front.tokenize
        self.loss_function = nn.CrossEntropyLoss()  # kept for attribute parity; the fused kernel computes the CE
        if config is not None:
            self.gather_embeddings = config.model.gather_embeddings
            self.in_batch_neg_num = config.data_config.in_batch_neg_num

    def zero_grad(self, set_to_none=False):
        """gradients live in the CLIP module's flat buffer: zero it (they are never detached to None)"""
        self.clip_model.zero_grad()

    def get_img_preprocess_fn(self):
        return self.img_preprocess_fn

    def get_tokenizer(self):
        def tokenizer_wrapper(txt):
            return self.tokenizer(txt, context_length=77, truncate=True)

        return tokenizer_wrapper

    def encode_text(self, text_tensor):
        return self.clip_model.encode_text(text_tensor)

    def encode_image(self, image_tensor):
        return self.clip_model.encode_image(image_tensor)

    def fuse_embeddings(self, img_emb, txt_emb):
        return img_emb + txt_emb

    # The reference runs BOTH towers on every item and multiplies by the modality masks (clip_sf.py:53-63); the collator feeds a
    # black image / an empty caption where a modality is absent (mbeir_dataset.py:427-434).  M-BEIR candidate pools are mostly
    # single-modality, and the image tower is 92 % of an item's FLOPs, so here each tower runs on the rows whose mask is 1 only
    # (gather -> tower -> scatter into zeros) and the same fused mask-and-add follows: x * 0 + y == y for finite x, so the result
    # is bitwise the dense one whenever the dense path's masked-out tower output is finite (tests/test_bench_paths_gpu.py).
    # Weight gradients lose the exact zeros of the masked rows, i.e. equal the dense ones up to fp32 summation order.
    compact_masked = True

    @staticmethod
    def _host_hint(mask):
        """the host copy of a modality mask that travelled with it (`mask._uniir_host`: host_utils.DevicePrefetcher /
        batch_to_device), or None.  Like the caption-length hint it is tied to the tensor's version: an in-place edit of the device
        mask after the hand-over invalidates it (the rows kept here must be the rows FuseFn's mask keeps)."""
        host = getattr(mask, "_uniir_host", None)
        ver = _tensor_version(mask)
        if not isinstance(host, torch.Tensor) or host.is_cuda or host.shape != mask.shape:
            return None
        if ver is None or getattr(mask, "_uniir_host_version", None) != ver:
            return None
        return host

    @classmethod
    def _live_rows(cls, *masks):
        """per mask: None if every row is live, else the int64 HOST index tensor of the live rows.  Masks without a valid host
        hint are read back together, in ONE device -> host transfer on the current stream (before the text leg forks)."""
        hosts = [cls._host_hint(m) for m in masks]
        missing = [i for i, h in enumerate(hosts) if h is None]
        if len(missing) == 1:
            hosts[missing[0]] = masks[missing[0]].detach().to("cpu")
        elif missing:
            if all(masks[i].shape == masks[missing[0]].shape and masks[i].dtype == masks[missing[0]].dtype for i in missing):
                both = torch.stack([masks[i].detach() for i in missing]).to("cpu")
                for j, i in enumerate(missing):
                    hosts[i] = both[j]
            else:
                for i in missing:
                    hosts[i] = masks[i].detach().to("cpu")
        out = []
        for host in hosts:
            live = torch.nonzero(host != 0).flatten()
            out.append(None if live.numel() == host.numel() else live)
        return out[0] if len(masks) == 1 else out

    def _encode_live(self, encode, inp, live_host):
        if live_host is None:
            return encode(inp)
        if live_host.numel() == 0 and torch.is_grad_enabled():
            # training: every rank must run every tower's backward (the overlapped gradient reducer announces a block's range from
            # it, and the ranks' collective sequences have to match) -- a batch with NO live row for this tower takes the dense path;
            # its outputs are multiplied by the zero masks as in the reference
            return encode(inp)
        live = live_host.to(inp.device, non_blocking=True)
        sub = inp.index_select(0, live)
        lens = getattr(inp, "_uniir_lens", None)                        # caption lengths travel with the rows (packed text tower)
        if isinstance(lens, torch.Tensor) and not lens.is_cuda and lens.dim() == 1 and lens.shape[0] == inp.shape[0] \
                and _tensor_version(inp) is not None and getattr(inp, "_uniir_lens_version", None) == _tensor_version(inp):
            sub._uniir_lens = lens[live_host]
            sub._uniir_lens_version = _tensor_version(sub)
        emb_live = encode(sub)                                          # [n_live, E]; n_live == 0 is legal
        full = torch.zeros(inp.shape[0], emb_live.shape[1], device=emb_live.device, dtype=emb_live.dtype)
        return full.index_copy(0, live, emb_live)                       # differentiable scatter: dead rows stay exact zeros

    def encode_multimodal_input(self, txt_tensor, img_tensor, txt_mask, img_mask):
        # the towers are independent up to the fusion: the text leg goes to the model's second stream (clip_model.CLIP.side_leg) and is
        # joined before FuseFn reads it; same kernels, same results.  Both masks' live rows are settled BEFORE the fork (a blocking
        # read inside the leg would stall the host while the image tower's launches wait behind it).
        if self.compact_masked:
            live_txt, live_img = self._live_rows(txt_mask, img_mask)
        with self.clip_model.side_leg(txt_tensor.device) as leg:
            if self.compact_masked:
                txt_emb = self._encode_live(self.encode_text, txt_tensor, live_txt)
            else:
                txt_emb = self.encode_text(txt_tensor)
        if self.compact_masked:
            img_emb = self._encode_live(self.encode_image, img_tensor, live_img)
        else:
            img_emb = self.encode_image(img_tensor)
        self.clip_model.join_leg(leg, txt_emb)
        return FuseFn.apply(txt_emb, img_emb, txt_mask, img_mask)  # txt*mask + img*mask, [batch, embed_dim]

    def get_logit_scale(self):
        return self.clip_model.logit_scale.exp()

    def compute_inbatch_contrastive_loss(self, batch):
        index_mapping = batch["index_mapping"]
        embeddings = self.encode_multimodal_input(batch["txt_batched"], batch["image_batched"],
                                                  batch["txt_mask_batched"], batch["image_mask_batched"])
        dev = embeddings.device
        idx_q = torch.tensor(index_mapping["query"], dtype=torch.int32).flatten().to(dev, non_blocking=True)
        idx_p = torch.tensor(index_mapping["pos_cand"], dtype=torch.int32).flatten().to(dev, non_blocking=True)
        if "neg_cand_list" in index_mapping:       # hard negatives: [bs, neg_num] rows of the flat batch
            idx_n = torch.tensor(index_mapping["neg_cand_list"], dtype=torch.int32).flatten().to(dev, non_blocking=True)
            loss, accuracy = HardNegNCEFn.apply(embeddings, idx_q, idx_p, idx_n, self.get_logit_scale(),
                                                int(getattr(self, "in_batch_neg_num", 0)))
            return {"loss": loss, "accuracy": accuracy}
        gather = bool(getattr(self, "gather_embeddings", False))
        loss, accuracy, _score = InBatchNCEFn.apply(embeddings, idx_q, idx_p, self.get_logit_scale(), gather)
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
