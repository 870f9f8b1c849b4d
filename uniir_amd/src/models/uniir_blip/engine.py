"""Drop-in module path for UniIR src/models/uniir_blip/engine.py (train_one_epoch, eval_engine: same signatures, the
alpha ramp config.model.alpha * min(1, i / len(loader)) in epoch 0 and during eval).  Loops: uniir_amd/host_utils.py.

eval_engine reproduces what the reference's save / restore really does: `state_dict()` aliases the live tensors, the
queues are reset with in-place copies (so they stay reset) and `_momentum_update` rebinds `param_m.data` (so the momentum
weights are what `load_state_dict(saved_state)` brings back).  Net effect after eval: online weights untouched, momentum
weights restored, queues / pointer left as eval wrote them.  Here the momentum weights are updated in place, so they are
snapshotted explicitly."""
import torch

from uniir_amd.host_utils import run_eval_epoch, run_train_epoch


def train_one_epoch(model, data_loader, optimizer, epoch, gpu_id, scheduler, global_step, scaler, config):
    full = config.model.alpha

    def step(m, batch, i, n):
        return m(batch=batch, alpha=full if epoch > 0 else full * min(1, i / n))

    return run_train_epoch(model, data_loader, optimizer, scheduler, config, gpu_id, epoch, step)


@torch.no_grad()
def eval_engine(model_without_ddp, model, data_loader, gpu_id, config):
    m = model_without_ddp
    m._sync()
    momentum_snapshot = m._mom.p32.clone()
    m.query_queue.copy_(torch.randn_like(m.query_queue))
    m.cand_queue.copy_(torch.randn_like(m.cand_queue))
    m.idx_queue.copy_(torch.full_like(m.idx_queue, -100))
    m.new_ptr_queue.zero_()
    m._ptr_host = 0
    print("Cleared model queue states.")
    stats = run_eval_epoch(model, data_loader, config, gpu_id,
                           lambda mm, batch, i, n: mm(batch=batch, alpha=config.model.alpha * min(1, i / n)))
    m._mom.p32.copy_(momentum_snapshot)
    m._mom.refresh_shadow()
    m._refresh_conv()
    print("Restored model queue states and model states from the saved variables.")
    return stats
