"""`FullyConnectedForecaster` (quantile_forecaster.py:4-170 of the reference): the frozen network the quantile policies invert
— for each of 19 probability points (0.05 .. 0.95) and each lead time it predicts cumulative demand over the next lead-time + 1
weeks from (past demands, days from christmas); `get_quantile` interpolates linearly between neighbouring points, with the 0-th
and 1-th quantiles extrapolated.  Same constructor, state-dict keys (`net.<i>.weight/bias`) and arithmetic — including the
float64 probability points that make the interpolation (and hence the orders) float64 upstream; the layers are `HipLinear`
(FP32 MFMA kernels).  Training the forecaster itself is out of scope (the reference ships it trained)."""
import datetime

import numpy as np
import torch
from torch import nn

from . import _lib
from .neural_networks import HipLazyLinear, _FusedELU


class FullyConnectedForecaster(nn.Module):
    def __init__(self, neurons_per_hidden_layer, lead_times, qs=np.arange(0.05, 1, 0.05), activation_function=None, device=None):
        super().__init__()
        self.qs = qs.round(2)
        self.qs_dict = {round(q, 2): i for i, q in enumerate(qs)}
        lead_times = torch.tensor(lead_times).int()
        self.lead_times = lead_times
        self.min_lead_time = min(lead_times)
        self.lead_times_dict = {lead_time: i for i, lead_time in enumerate(lead_times)}
        self.device = torch.device(device) if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.prob_points = torch.tensor([0] + list(self.qs) + [1]).to(self.device)  # float64, as upstream
        self.name = datetime.datetime.now().strftime("%Y-%m-%d_%H-%M-%S")
        self.layers = []
        fuse = activation_function is None or isinstance(activation_function, nn.ELU)
        for width in neurons_per_hidden_layer:
            lin = HipLazyLinear(width)
            if fuse:
                lin.fused_act = _lib.NIC_ACT_ELU
                self.layers += [lin, _FusedELU()]
            else:
                self.layers += [lin, activation_function]
        self.layers.append(HipLazyLinear(len(qs) * len(self.lead_times)))
        self.net = nn.Sequential(*self.layers)

    def forward(self, x):
        x = self.net(x)
        return torch.clip(x, min=0).reshape(*x.shape[:-1], len(self.qs), len(self.lead_times))

    def get_quantile(self, x, quantile, lead_times):
        """x (B, S, features); quantile (B, S) in (0, 1); lead_times (B, S).  quantile_forecaster.py:62-106."""
        indices = torch.searchsorted(self.prob_points, quantile.detach().contiguous())
        x = self.forward(x)
        x = self.retrieve_corresponding_lead_time(x, lead_times)
        x = self.create_0_1_quantiles(x)
        prev_quantile = torch.gather(x, 2, (indices - 1).unsqueeze(2)).squeeze(2)
        next_quantile = torch.gather(x, 2, indices.unsqueeze(2)).squeeze(2)
        diff_prev = quantile - self.prob_points[indices - 1]
        diff_next = self.prob_points[indices] - quantile
        return prev_quantile + (next_quantile - prev_quantile) * diff_prev / (diff_prev + diff_next)

    def create_0_1_quantiles(self, x):
        return torch.cat([(2 * x[:, :, 0] - x[:, :, 1]).unsqueeze(2), x, (2 * x[:, :, -1] - x[:, :, -2]).unsqueeze(2)], dim=2)

    def retrieve_corresponding_lead_time(self, x, lead_times):
        dif = (lead_times - self.min_lead_time).to(torch.int64)
        return torch.gather(x, 3, dif.unsqueeze(2).expand(-1, -1, x.shape[2]).unsqueeze(3)).squeeze(3)
