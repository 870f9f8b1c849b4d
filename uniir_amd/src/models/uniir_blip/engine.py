"""Train / eval epoch loops for the BLIP models (drop-in for UniIR src/models/uniir_blip/engine.py:
train_one_epoch :9-66, eval_engine :69-114).  Same signatures, alpha ramp (epoch 0: alpha * min(1, i / len(loader))),
gradient accumulation and logging.

Differences on MI355X (result-preserving): no autocast / GradScaler (bf16 MFMA compute, fp32 loss), the optimizer step is
the fused AdamW over the flat buffer with one RCCL all-reduce (uniir_amd.trainer.NativeAdamW).

eval_engine reproduces what the reference's save/restore really does: `state_dict()` aliases the live tensors, the
queues are reset with in-place copies (so they stay reset) and `_momentum_update` rebinds `param_m.data` (so the
momentum weights are what `load_state_dict(saved_state)` brings back).  Net effect after eval: online weights untouched,
momentum weights restored, queues / pointer left as eval wrote them.  Here the momentum weights are updated in place, so
they are snapshotted explicitly.
"""
import torch

from models.uniir_blip import utils


def _to_device(batch, gpu_id):
    for key, value in batch.items():
        if isinstance(value, torch.Tensor):
            batch[key] = value.to(gpu_id, non_blocking=True)
        elif hasattr(value, "items") and not isinstance(value, dict):      # transformers BatchEncoding
            for k, v in value.items():
                value[k] = v.to(gpu_id)
    return batch


def train_one_epoch(model, data_loader, optimizer, epoch, gpu_id, scheduler, global_step, scaler, config):
    model.train()
    logger = utils.MetricLogger(delimiter="  ")
    logger.add_meter("lr", utils.SmoothedValue(window_size=1, fmt="{value:.6f}"))
    logger.add_meter("loss", utils.SmoothedValue(window_size=1, fmt="{value:.4f}"))
    logger.add_meter("inbatch_accuracy", utils.SmoothedValue(window_size=1, fmt="{value:.4f}"))
    accumulation_steps = config.trainer_config.gradient_accumulation_steps
    pending = 0
    n = len(data_loader)
    for i, batch in enumerate(logger.log_every(data_loader, config.trainer_config.print_freq, f"Train Epoch: [{epoch}]")):
        batch = _to_device(batch, gpu_id)
        alpha = config.model.alpha if epoch > 0 else config.model.alpha * min(1, i / n)
        outputs = model(batch=batch, alpha=alpha)
        loss = outputs["loss"] / accumulation_steps
        loss.backward()
        pending += 1
        if pending == accumulation_steps:
            global_step += 1
            optimizer.step()
            model.zero_grad()
            scheduler.step()
            pending = 0
        logger.update(loss=loss.item() * accumulation_steps)
        logger.update(lr=optimizer.param_groups[0]["lr"])
        logger.update(inbatch_accuracy=outputs["accuracy"].item())
    logger.synchronize_between_processes()
    print("Averaged stats:", logger.global_avg())
    return {k: meter.global_avg for k, meter in logger.meters.items()}


@torch.no_grad()
def eval_engine(model_without_ddp, model, data_loader, gpu_id, config):
    model.eval()
    logger = utils.MetricLogger(delimiter="  ")
    logger.add_meter("loss", utils.SmoothedValue(window_size=1, fmt="{value:.4f}"))
    logger.add_meter("inbatch_accuracy", utils.SmoothedValue(window_size=1, fmt="{value:.4f}"))
    m = model_without_ddp
    m._sync()
    momentum_snapshot = m._mom.p32.clone()
    m.query_queue.copy_(torch.randn_like(m.query_queue))
    m.cand_queue.copy_(torch.randn_like(m.cand_queue))
    m.idx_queue.copy_(torch.full_like(m.idx_queue, -100))
    m.new_ptr_queue.zero_()
    m._ptr_host = 0
    print("Cleared model queue states.")
    n = len(data_loader)
    for i, batch in enumerate(logger.log_every(data_loader, config.evaluator.print_freq, "Test:")):
        batch = _to_device(batch, gpu_id)
        outputs = model(batch=batch, alpha=config.model.alpha * min(1, i / n))
        logger.update(loss=outputs["loss"].item())
        logger.update(inbatch_accuracy=outputs["accuracy"].item())
    logger.synchronize_between_processes()
    print("Averaged stats:", logger.global_avg())
    m._mom.p32.copy_(momentum_snapshot)
    m._mom.refresh_shadow()
    m._refresh_conv()
    print("Restored model queue states and model states from the saved variables.")
    return {k: meter.global_avg for k, meter in logger.meters.items()}
