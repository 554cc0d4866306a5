"""apex.optimizers.FusedAdam(params, lr, bias_correction=True, betas, eps, adam_w_mode=True, weight_decay, amsgrad=False,
set_grad_none=True) on permuto_sdf_amd.optim.FusedAdamW: the decoupled-weight-decay Adam of apex's default `adam_w_mode`, which
is the update of torch.optim.AdamW (the reference's other branch, train_permuto_sdf.py:303).  One launch per large tensor, one
per 64 small ones, no per-parameter host work (torch's multi-tensor Adam spends ~1.9 ms of host time per step on the reference's
five parameter groups: profiles/r03_reference_cprofile.txt)."""
from permuto_sdf_amd.optim import FusedAdamW


class FusedAdam(FusedAdamW):
    def __init__(self, params, lr=1e-3, bias_correction=True, betas=(0.9, 0.999), eps=1e-8, adam_w_mode=True, weight_decay=0.0,
                 amsgrad=False, set_grad_none=True, capturable=False, master_weights=False):
        if amsgrad:
            raise RuntimeError("FusedAdam does not support the AMSGrad variant.")      # as apex
        if not bias_correction or not adam_w_mode or capturable or master_weights:
            raise NotImplementedError("compat apex.optimizers.FusedAdam: only bias_correction=True, adam_w_mode=True, "
                                      "capturable=False, master_weights=False (what the reference uses)")
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.set_grad_none = bool(set_grad_none)

    def zero_grad(self, set_to_none=None):
        super().zero_grad(set_to_none=self.set_grad_none if set_to_none is None else set_to_none)
