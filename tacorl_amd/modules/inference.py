"""Rollout-time surface of the module classes (reference evaluation/rollout_manager.py:310-431):
`module.actor.get_actions(obs, deterministic, reparameterize)` (actor.py:65-111 through
visual_actor_wrapper.py:41-76), `module.perceptual_encoder.get_state_from_observation(observation, modalities)`
(representation_network.py:36-71), `module.action_decoder.act / clear_hidden_state`
(action_decoder_logistic.py:73-97).  Small-batch latency path on the library's per-layer kernels (any number of
images, any camera geometry); nothing here is captured in a graph or keeps gradients.
"""
import torch

from .. import ops
from .._lib import BF16, F32, call, ptr


class EncoderRunner:
    """LMPVisionEncoder forward of N images through one network block (reference encoder.py:369-419)."""

    def __init__(self, owner):
        self.owner = owner
        self._buf = {}

    def _buffers(self, n, hw):
        key = (n, hw)
        if key not in self._buf:
            o = self.owner
            ops.note_alloc()
            self._buf[key] = (torch.zeros(n, *hw, 3, device=o.dev, dtype=o.img_dtype), torch.zeros(n, 32, device=o.dev),
                              torch.zeros(ops.encoder_act_layout(n, *hw)[1], device=o.dev))
        return self._buf[key]

    def encode(self, enc_ptr, images):
        """images (N,3,H,W) or (3,H,W) fp32 in [-1,1] (the transformed observation) -> (N,32) embeddings."""
        o = self.owner
        x = images.to(o.dev, torch.float32)
        if x.dim() == 3:
            x = x.unsqueeze(0)
        x = x.contiguous()
        n, _, H, W = x.shape
        img, out, act = self._buffers(n, (H, W))
        xd = BF16 if o.img_dtype == torch.bfloat16 else F32
        call("tacorl_pack_images", ptr(x), 3 * H * W, 1, ptr(img), xd, n, 3, H, W, ops.stream())
        call("tacorl_encoder_fwd", 1, ops.ptr_array([img]), ops.ptr_array([enc_ptr]), ops.ptr_array([out]),
             ops.ptr_array([act]), ops.int_array([n]), H, W, xd, o.compute, ops.stream())
        # a copy, not the cached buffer: two cameras of one geometry (or obs and goal) share (n, hw) and hence `out`
        return out.clone()


def state_from_observation(owner, runner, net, observation, modalities):
    """LateFusion.get_state_from_observation: per-camera embeddings concatenated in `modalities` order."""
    embs = [runner.encode(net.enc(c), observation[c]) for c in modalities]
    return embs[0] if len(embs) == 1 else torch.cat(embs, dim=-1)


class ActorSurface:
    """Bound onto `module.actor`: get_actions of VisualActorWrapper + Actor + MLPPolicy for one network block."""

    def __init__(self, owner, net, cams, goal_cams, action_dim, discrete_gripper):
        self.owner, self.net, self.cams, self.goal_cams = owner, net, list(cams), list(goal_cams)
        self.A, self.dg = action_dim, bool(discrete_gripper)
        self.Ac = action_dim - 1 if self.dg else action_dim
        self.HD = 2 * self.Ac + (2 if self.dg else 0)
        self.runner = EncoderRunner(owner)
        self._buf = {}

    def emb_representation(self, obs):
        """[enc(obs) | goal_encoder(enc(goal))] (visual_actor_wrapper.py:41-62); a tensor passes through."""
        if not isinstance(obs, dict):
            return obs.to(self.owner.dev, torch.float32)
        o, net = self.owner, self.net
        if "goal" not in obs:
            return state_from_observation(o, self.runner, net, obs.get("observation", obs), self.cams)
        e_obs = state_from_observation(o, self.runner, net, obs["observation"], self.cams)
        e_goal = state_from_observation(o, self.runner, net, obs["goal"], self.goal_cams)
        n = e_obs.shape[0]
        gact = torch.zeros(ops.mlp_act_layout(n, net.genc_dims, net.genc_acts)[2], device=o.dev)
        ops.mlp_fwd([e_goal], e_goal.shape[1], [net.genc()], [gact], [n], net.genc_dims, net.genc_acts, o.compute)
        yo = ops.mlp_act_layout(n, net.genc_dims, net.genc_acts)[1][-1]
        return torch.cat([e_obs, gact[yo: yo + n * net.G].view(n, net.G)], dim=-1)

    def get_actions(self, observation, deterministic=False, reparameterize=False, noise=None):
        """actor.py:65-111.  Returns (actions (N,A), log_pi): log_pi is zeros_like(actions) when deterministic, (N,1)
        otherwise.  noise: {'eps': (N,Ac) N(0,1)[, 'gumbel_u': (N,2) U(0,1)]} injects the draws."""
        o, net = self.owner, self.net
        s = self.emb_representation(observation).contiguous()
        n = s.shape[0]
        pact = torch.zeros(ops.mlp_act_layout(n, net.head_dims, net.head_acts)[2], device=o.dev)
        ops.mlp_fwd([s], s.shape[1], [net.head()], [pact], [n], net.head_dims, net.head_acts, o.compute)
        yo = ops.mlp_act_layout(n, net.head_dims, net.head_acts)[1][-1]
        head = pact[yo: yo + n * self.HD]
        f = lambda *sh: torch.zeros(*sh, device=o.dev)  # noqa: E731
        act, logp = f(n, self.A), f(n)
        if deterministic:
            # tanh(mean) and the argmax gripper class = the sampling kernel with eps = 0 and equal Gumbel noise
            eps, gu = f(n, self.Ac), torch.full((n, 2), 0.5, device=o.dev) if self.dg else None
        else:
            eps = noise["eps"].to(o.dev).reshape(n, self.Ac).contiguous() if noise else torch.randn(n, self.Ac, device=o.dev)
            gu = None
            if self.dg:
                gu = noise["gumbel_u"].to(o.dev).reshape(n, 2).contiguous() if noise else torch.rand(n, 2, device=o.dev)
        grip = torch.zeros(n, dtype=torch.int32, device=o.dev) if self.dg else None
        ops.tanh_normal_sample(head, self.HD, eps, gu, bool(reparameterize and not deterministic), act, 0, self.A, logp, grip,
                               1, n, self.Ac)
        if deterministic:
            return act, torch.zeros_like(act)
        return act, logp.view(n, 1)


def attach_rollout_surface(module, actor_net, cams, goal_cams, action_dim, discrete_gripper, lmp_net=None, lmp_cams=None,
                           ad=None):
    """Give the module's reference-named containers their rollout methods."""
    surf = ActorSurface(module, actor_net, cams, goal_cams, action_dim, discrete_gripper)
    module.__dict__["_actor_surface"] = surf
    module.actor.get_actions = surf.get_actions
    module.actor.get_emb_representation = surf.emb_representation
    module.actor.action_dim = action_dim
    module.actor.discrete_gripper = bool(discrete_gripper)
    if lmp_net is not None:
        runner = EncoderRunner(module)
        module.__dict__["_pe_runner"] = runner
        module.perceptual_encoder.get_state_from_observation = (
            lambda observation, modalities=None: state_from_observation(module, runner, lmp_net, observation,
                                                                        list(modalities or lmp_cams)))
    if ad is not None:
        module.action_decoder.clear_hidden_state = ad.clear_hidden_state
        module.action_decoder.act = lambda latent_plan, perceptual_emb, latent_goal=None, noise=None: ad.act(
            latent_plan, perceptual_emb, latent_goal, noise=noise, compute=module.compute)
