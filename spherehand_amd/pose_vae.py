"""Frozen VAE pose prior -- the reference's network/pose_vae.py (PoseVae :11-99), state-dict
compatible (`base.*`, `mu.*`, `logvar.*`, `decoder.*`).  MultiTaskLoss's `pose_prior` term
is `1e-2 * prior_loss(xyz / 100)` per stack (network/create_network_and_criterion.py:164,
:238-243): a 123-256-256-32-256-256-123 MLP on B*V samples -- torch ops, not a kernel of
this path.

Weights: ``spherehand_amd/data/pose_vae.npz`` is a plain re-export (data only,
tests/golden/make_goldens_priors.py) of the reference's mesh/model/pose_vae.pth.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .pose_denoiser import _DATA, load_npz_state_dict

DEFAULT_WEIGHTS = os.path.join(_DATA, "pose_vae.npz")


def _mlp(n_in, n_out=None):
    layers = [nn.Linear(n_in, 256), nn.GroupNorm(16, 256), nn.ReLU(),
              nn.Linear(256, 256), nn.GroupNorm(16, 256), nn.ReLU()]
    if n_out is not None:
        layers.append(nn.Linear(256, n_out))
    return nn.Sequential(*layers)


class PoseVae(nn.Module):
    def __init__(self, pose_fea, latent_fea, model_path=None):
        super().__init__()
        self.pose_fea, self.latent_fea = pose_fea, latent_fea
        self.base = _mlp(pose_fea)
        self.mu = nn.Linear(256, latent_fea)
        self.logvar = nn.Linear(256, latent_fea)
        self.decoder = _mlp(latent_fea, pose_fea)
        if model_path is not None:
            if str(model_path).endswith('.npz'):
                load_npz_state_dict(self, model_path)
            else:
                self.load_state_dict(torch.load(model_path, map_location='cpu')['network_state_dict'])
                for p in self.parameters():
                    p.requires_grad = False

    def _reparameterize(self, mu, logvar, eps=None):
        std = torch.exp(0.5 * logvar) * 0.1
        if eps is None:
            eps = torch.randn_like(std)
        return eps.mul(std).add_(mu)

    def _likelihood(self, x, recon_x, mu, logvar):
        # reconstruction (mean) + KL (sum), the reference's mix of reductions (:52-59)
        return F.mse_loss(x, recon_x) - 0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp())

    def forward(self, x, do_reparameterize=False):
        base = self.base(x)
        mu, logvar = self.mu(base), self.logvar(base)
        z = self._reparameterize(mu, logvar) if do_reparameterize else mu
        recon_x = self.decoder(z)
        return recon_x, mu, logvar, self._likelihood(x, recon_x, mu, logvar)

    def prior_loss(self, x, eps=None):
        """x [..., 41, 3] in units of 100 mm -> scalar.  `eps` (optional, [N,latent]) replaces the
        reparameterisation draw (tests; the reference always draws, :84-88)."""
        x = x.reshape(-1, self.pose_fea)
        base = self.base(x)
        mu, logvar = self.mu(base), self.logvar(base)
        recon_x = self.decoder(self._reparameterize(mu, logvar, eps))
        return self._likelihood(x, recon_x, mu, logvar)

    def recons(self, x):
        num_batch, num_view = x.shape[0], x.shape[1]
        mu = self.mu(self.base(x.reshape(-1, self.pose_fea)))
        return self.decoder(mu).view(num_batch, num_view, -1, 3)


def default_pose_vae():
    """The prior MultiTaskLoss builds when `--prior` is on (PoseVae(41*3, 32, 'mesh/model/pose_vae.pth'))."""
    if not os.path.exists(DEFAULT_WEIGHTS):
        raise FileNotFoundError('skipped: asset missing ({})'.format(DEFAULT_WEIGHTS))
    return PoseVae(41 * 3, 32, model_path=DEFAULT_WEIGHTS)
