"""DDIM eta.  Mirrors ``dppo/model/diffusion/eta.py:12-40`` (reference ``EtaFixed``); learned etas are out of scope."""
import torch


class EtaFixed(torch.nn.Module):
    def __init__(self, base_eta=0.5, min_eta=0.1, max_eta=1.0, **kwargs):
        super().__init__()
        self.eta_logit = torch.nn.Parameter(torch.ones(1))
        self.min, self.max = min_eta, max_eta
        self.eta_logit.data = torch.atanh(torch.tensor([2 * (base_eta - min_eta) / (max_eta - min_eta) - 1]))

    def value(self) -> float:
        """The scalar the reference broadcasts with .item() (eta.py:36-40)."""
        eta = 0.5 * (torch.tanh(self.eta_logit.detach().float().cpu()) + 1) * (self.max - self.min) + self.min
        return eta.item()

    def __call__(self, cond):
        data = cond["state"] if "state" in cond else cond["rgb"]
        return torch.full((len(data), 1), self.value()).to(data.device)
