"""Drop-in module path for UniIR src/models/uniir_clip/engine.py (train_one_epoch, eval_engine: same signatures).
The loops are uniir_amd/host_utils.py's; on MI355X there is no autocast / GradScaler arithmetic (bf16 MFMA towers,
fp32 loss; `scaler` is accepted and ignored) and the optimizer step is the fused AdamW with one RCCL all-reduce."""
from uniir_amd.host_utils import run_eval_epoch, run_train_epoch


def _step(model, batch, i, n):
    return model(batch)


def train_one_epoch(model, data_loader, optimizer, epoch, gpu_id, scheduler, global_step, scaler, config):
    return run_train_epoch(model, data_loader, optimizer, scheduler, config, gpu_id, epoch, _step)


def eval_engine(model, data_loader, gpu_id, config):
    return run_eval_epoch(model, data_loader, config, gpu_id, _step)
