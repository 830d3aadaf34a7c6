"""Predictor head (reference models_mae/MLP.py:4-10): Linear -> BatchNorm1d(channel = token position) -> ReLU -> Linear.
A plain nn.Sequential so `predictor.{0,1,3}.*` keys and torch-default init match; compute runs in libcsmae_hip."""
import torch.nn as nn


def MLP(emd_dim, channel=64, hidden_size=1024):
    return nn.Sequential(nn.Linear(emd_dim, hidden_size), nn.BatchNorm1d(channel), nn.ReLU(inplace=True), nn.Linear(hidden_size, emd_dim))
