"""Decision-transformer policy that hands (T, sigma_d, mu) to the HIP ADMM loop each step.

Stays in PyTorch-ROCm (BASELINE.json north_star): ~1.3 M parameters, <= 18 tokens of context.  Written fresh
against the behaviour of /root/reference/transformer/decision_transformer.py with the SAME parameter names, so a
checkpoint trained with the reference (`checkpoints/model_experiment_{1,2}.pt`, eval.py:20,27) loads unchanged:

    time_embed, task_embed, embed_action.0, embed_return.0, layer_n, state_encoder.{0,2,4,7},
    transformer.<i>.{ln1, c_att.{qkv_proj,o_proj,masking}, ln2, mlp.{fc,fc_proj}}, predict_action.0, predict_rtg

Reference behaviours kept on purpose (decision_transformer.py):
  * a block is `x = x + attn(ln1(x)); x = mlp(ln2(x))` - the MLP branch has NO residual (:99-102)
  * the state encoder is hard-wired to 128 x 128 (Linear(2304, .), :128-132); larger slices are area-averaged
    down to 128 x 128 by `policy_observation` (documented deviation: the reference cannot run them at all)
  * action order / scaling: norm = (T, sigma_d*70/255, mu), flex = (mu, sigma_d*70/255, T)  (:147-154, :266-275)
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F


class DecisionTransformerConfig:
    """Same attribute bag as the reference (decision_transformer.py:279-291)."""
    dropout = 0.1
    embd_dropout = 0.1
    embed_dim = 128
    n_heads = 4
    action_dim = 3
    max_timestep = 30
    n_blocks = 5
    block_size = 18
    n_embeds = 9
    mode = "norm"

    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)


FUSED_ATTENTION = True     # GPU: F.scaled_dot_product_attention (tests switch it off to compare the two forms)


class _Attention(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.qkv_proj = nn.Linear(cfg.embed_dim, 3 * cfg.embed_dim)
        self.o_proj = nn.Linear(cfg.embed_dim, cfg.embed_dim)
        self.n_heads = cfg.n_heads
        # kept as a buffer under the reference's name so state_dicts match key-for-key
        self.register_buffer("masking", torch.tril(torch.ones(cfg.block_size, cfg.block_size))
                             .view(1, 1, cfg.block_size, cfg.block_size))

    def forward(self, x):
        b, t, e = x.shape
        q, k, v = self.qkv_proj(x).view(b, t, 3, self.n_heads, e // self.n_heads).permute(2, 0, 3, 1, 4)
        if x.is_cuda and FUSED_ATTENTION:
            # one fused kernel instead of five (scores, scale, mask, softmax, weighted sum): the policy side of a DT-driven step
            # is ~110 kernels of a few microseconds each, a fifth of them here.  Same arithmetic up to the softmax's summation order.
            y = F.scaled_dot_product_attention(q, k, v, is_causal=True)
        else:
            att = (q @ k.transpose(-1, -2)) / math.sqrt(q.size(-1))
            att = att.masked_fill(self.masking[..., :t, :t] == 0, float("-inf"))
            y = F.softmax(att, dim=-1) @ v
        return self.o_proj(y.transpose(1, 2).reshape(b, t, e))


class _MLP(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.fc = nn.Linear(cfg.embed_dim, 4 * cfg.embed_dim)
        self.fc_proj = nn.Linear(4 * cfg.embed_dim, cfg.embed_dim)

    def forward(self, x):
        return self.fc_proj(F.gelu(self.fc(x)))


class _Block(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.ln1 = nn.LayerNorm(cfg.embed_dim)
        self.c_att = _Attention(cfg)
        self.ln2 = nn.LayerNorm(cfg.embed_dim)
        self.mlp = _MLP(cfg)

    def forward(self, x):
        x = x + self.c_att(self.ln1(x))
        return self.mlp(self.ln2(x))          # no residual on the MLP branch (reference :101)


def policy_observation(x: torch.Tensor) -> torch.Tensor:
    """[N,1,H,W] (or [N,H*W]) real image -> [N, 128*128] policy observation (env.get_policy_ob layout)."""
    if x.dim() == 2:
        side = int(round(math.sqrt(x.shape[1])))
        x = x.reshape(x.shape[0], 1, side, side)
    if x.shape[-2:] != (128, 128):
        x = F.adaptive_avg_pool2d(x, (128, 128))
    return x.reshape(x.shape[0], -1)


class DecisionTransformer(nn.Module):
    """Inference-only (dropout layers of the reference are identities in eval mode and carry no parameters)."""

    def __init__(self, config) -> None:
        super().__init__()
        e = config.embed_dim
        self.action_dim = config.action_dim
        self.embed_dim = e
        self.time_embed = nn.Embedding(config.max_timestep, e)
        self.task_embed = nn.Embedding(config.n_embeds, e)
        self.embed_action = nn.Sequential(nn.Linear(self.action_dim, e), nn.Tanh())
        self.embed_return = nn.Sequential(nn.Linear(1, e), nn.Tanh())
        self.layer_n = nn.LayerNorm(e)
        self.state_encoder = nn.Sequential(
            nn.Conv2d(1, 8, 8, stride=4), nn.ReLU(), nn.Conv2d(8, 16, 4, stride=2), nn.ReLU(),
            nn.Conv2d(16, 16, 3, stride=1), nn.ReLU(), nn.Flatten(), nn.Linear(2304, e), nn.Tanh())
        self.transformer = nn.Sequential(*[_Block(config) for _ in range(config.n_blocks)])
        self.predict_action = nn.Sequential(nn.Linear(e, self.action_dim), nn.Sigmoid())
        self.predict_rtg = nn.Linear(e, 1)
        order = ("mu", "sigma_d", "T") if getattr(config, "mode", "norm") == "flex" else ("T", "sigma_d", "mu")
        scale = {"mu": 1.0, "sigma_d": 70.0 / 255.0, "T": 1.0}
        self.action_range = OrderedDict((k, {"scale": scale[k], "shift": 0.0}) for k in order)
        self.eval()

    def _transform_actions(self, outputs):
        parts = torch.split(outputs, outputs.shape[-1] // self.action_dim, dim=-1)
        action_dict = OrderedDict()
        for part, (key, rng) in zip(parts, self.action_range.items()):
            action_dict[key] = part * rng["scale"] + rng["shift"]
        return torch.cat(list(action_dict.values()), dim=-1), action_dict

    def encode_states(self, states: torch.Tensor) -> torch.Tensor:
        """[..., 16384] observations -> [..., E] state-encoder outputs (before the task embedding is added).  The encoder
        is a pure per-image function, so a driver may run it ONCE per observation and hand the cached result to `forward`
        (`state_emb=`) instead of re-encoding the whole context window in both forwards of every step as the reference does
        (eval.py:150-186: 2 x 6 images per step and slice)."""
        lead = states.shape[:-1]
        return self.state_encoder(states.reshape(-1, 1, 128, 128)).reshape(*lead, -1)

    def forward(self, rtg, states, timesteps, task, actions: Optional[torch.Tensor] = None,
                eval_rtg: bool = False, eval_actions: bool = False, state_emb: Optional[torch.Tensor] = None):
        """rtg [B,T,1]; states [B,T,16384]; timesteps [B,T,1]; task [B,T]; actions [B,T,3] or None.
        Token order per step: (rtg, state, action) or (rtg, state) when actions is None (:212-263).
        state_emb [B,T,E]: `encode_states(states)` computed earlier (then `states` is only read for its shape)."""
        b, t = states.shape[0], states.shape[1]
        rtg_e = self.embed_return(rtg)
        st_e = self.encode_states(states) if state_emb is None else state_emb
        time_e = self.time_embed(timesteps.to(torch.int64).reshape(b, -1))
        st_e = st_e + self.task_embed(task)
        per = 3 if actions is not None else 2
        tok = torch.zeros((b, per * t, self.embed_dim), device=st_e.device, dtype=st_e.dtype)
        tok[:, 0::per] = rtg_e
        tok[:, 1::per] = st_e
        if actions is not None:
            tok[:, 2::per] = self.embed_action(actions)
        x = self.layer_n(self.transformer(tok + torch.repeat_interleave(time_e, per, dim=1)))
        pred_actions, action_dict = self._transform_actions(self.predict_action(x[:, 1::per]))
        if eval_rtg:
            return self.predict_rtg(x[:, 2::3])
        if eval_actions or actions is None:
            return pred_actions, action_dict
        return torch.cat([pred_actions, self.predict_rtg(x[:, 2::3])], dim=-1), action_dict
