"""Train / eval epoch loops (drop-in for UniIR src/models/uniir_clip/engine.py: train_one_epoch :7-55,
eval_engine :58-84).

Same call signature and logging.  What differs on MI355X, without changing results:
  * no autocast / GradScaler arithmetic: the towers compute in bf16 inside libuniir_hip.so and bf16 needs no loss
    scaling (`scaler` is accepted and ignored when it is None or disabled);
  * when `optimizer` is a uniir_amd NativeOptimizer (train.py builds one) the optimizer step is the fused AdamW over
    the flat parameter buffer and the gradient all-reduce is one RCCL call; a plain torch optimizer also works.
"""
import torch

from models.uniir_clip import utils


def _to_device(batch, gpu_id):
    for key in batch:
        if isinstance(batch[key], torch.Tensor):
            batch[key] = batch[key].to(gpu_id, non_blocking=True)
    return batch


def train_one_epoch(model, data_loader, optimizer, epoch, gpu_id, scheduler, global_step, scaler, config):
    model.train()
    logger = utils.MetricLogger(delimiter="  ")
    logger.add_meter("lr", utils.SmoothedValue(window_size=1, fmt="{value:.6f}"))
    logger.add_meter("loss", utils.SmoothedValue(window_size=1, fmt="{value:.4f}"))
    logger.add_meter("inbatch_accuracy", utils.SmoothedValue(window_size=1, fmt="{value:.4f}"))
    accumulation_steps = config.trainer_config.gradient_accumulation_steps
    pending = 0
    for batch in logger.log_every(data_loader, config.trainer_config.print_freq, f"Train Epoch: [{epoch}]"):
        batch = _to_device(batch, gpu_id)
        outputs = model(batch)
        loss = outputs["loss"] / accumulation_steps      # backward averages over the accumulation window
        loss.backward()
        pending += 1
        if pending == accumulation_steps:
            global_step += 1
            optimizer.step()
            model.zero_grad()
            scheduler.step()
            pending = 0
        logger.update(loss=loss.item() * accumulation_steps)      # host sync, as in the reference (engine.py:48)
        logger.update(lr=optimizer.param_groups[0]["lr"])
        logger.update(inbatch_accuracy=outputs["accuracy"].item())
    logger.synchronize_between_processes()
    print("Averaged stats:", logger.global_avg())
    return {k: meter.global_avg for k, meter in logger.meters.items()}


@torch.no_grad()
def eval_engine(model, data_loader, gpu_id, config):
    model.eval()
    logger = utils.MetricLogger(delimiter="  ")
    logger.add_meter("loss", utils.SmoothedValue(window_size=1, fmt="{value:.4f}"))
    logger.add_meter("inbatch_accuracy", utils.SmoothedValue(window_size=1, fmt="{value:.4f}"))
    for batch in logger.log_every(data_loader, config.evaluator.print_freq, "Test:"):
        batch = _to_device(batch, gpu_id)
        outputs = model(batch)
        logger.update(loss=outputs["loss"].item())
        logger.update(inbatch_accuracy=outputs["accuracy"].item())
    logger.synchronize_between_processes()
    print("Averaged stats:", logger.global_avg())
    return {k: meter.global_avg for k, meter in logger.meters.items()}
