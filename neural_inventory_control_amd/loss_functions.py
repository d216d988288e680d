"""`PolicyLoss` (loss_functions.py:3-12 of the reference): the loss IS the summed cost."""
from torch import nn


class PolicyLoss(nn.Module):
    def forward(self, observation, action, reward):
        return reward.sum()
