"""DDIM eta.  Mirrors ``dppo/model/diffusion/eta.py:12-40`` (reference ``EtaFixed``); learned etas are out of scope."""
import torch


class EtaFixed(torch.nn.Module):
    def __init__(self, base_eta=0.5, min_eta=0.1, max_eta=1.0, **kwargs):
        super().__init__()
        self.eta_logit = torch.nn.Parameter(torch.ones(1))
        self.min, self.max = min_eta, max_eta
        self.eta_logit.data = torch.atanh(torch.tensor([2 * (base_eta - min_eta) / (max_eta - min_eta) - 1]))

    def value(self) -> float:
        """The scalar the reference broadcasts with .item() (eta.py:36-40).  The logit is frozen (learned etas are
        rejected), so the value is read back from the device once per write of the parameter, not once per sampler call
        (a D2H sync per call would serialise the pipelined rollout)."""
        key = (self.eta_logit.data_ptr(), self.eta_logit._version)
        hit = self.__dict__.get("_value_cache")
        if hit is None or hit[0] != key:
            eta = 0.5 * (torch.tanh(self.eta_logit.detach().float().cpu()) + 1) * (self.max - self.min) + self.min
            hit = (key, eta.item())
            self.__dict__["_value_cache"] = hit
        return hit[1]

    def __call__(self, cond):
        data = cond["state"] if "state" in cond else cond["rgb"]
        return torch.full((len(data), 1), self.value()).to(data.device)
