"""Train-time image augmentations as GPU work (reference utils/transforms.py:265-330 and
config/datamodule/transform_manager/transforms/rl_train.yaml): the random draws are made here, as small device
tables, and applied by `tacorl_pack_images_u8_aug_batch` on the way into the encoder's image buffers."""
import torch


class AugmentSpec:
    """One camera's train pipeline: RandomShiftsAug(pad) and ColorTransform(contrast, brightness, hue, prob)
    (rl_train.yaml: pad 6 / 4, contrast 0.1, brightness 0.1, hue 0.02)."""

    def __init__(self, pad=4, brightness=0.1, contrast=0.1, hue=0.02, prob=1.0, resize=None):
        self.pad, self.brightness, self.contrast, self.hue, self.prob = int(pad), brightness, contrast, hue, prob
        # torchvision.transforms.Resize(size) ahead of the shift (rl_train.yaml:3-4,16-17: [128, 128] static, [84, 84]
        # gripper): the dataset's frames keep their camera resolution and are resized by the same launch
        self.resize = None if resize is None else (int(resize[0]), int(resize[1]))

    def draw(self, n, device, generator=None):
        """The reference's draws for n frames: shift = randint(0, 2*pad+1, (n,2)) (utils/transforms.py:288-290);
        ColorJitter.get_params: fn_idx = randperm(4), b ~ U(max(0,1-b), 1+b), c likewise, h ~ U(-hue, hue);
        ColorTransform applies it with probability `prob` (:311-314).  Returns (shift int32 (n,2), jitter f32 (n,8))."""
        g = generator
        shift = torch.randint(0, 2 * self.pad + 1, (n, 2), device=device, generator=g, dtype=torch.int32)
        u = torch.rand(n, 4, device=device, generator=g)
        b = max(0.0, 1 - self.brightness) + u[:, 0] * (1 + self.brightness - max(0.0, 1 - self.brightness))
        c = max(0.0, 1 - self.contrast) + u[:, 1] * (1 + self.contrast - max(0.0, 1 - self.contrast))
        h = (2 * u[:, 2] - 1) * self.hue
        order = torch.rand(n, 4, device=device, generator=g).argsort(dim=1).float()  # a uniform random permutation
        apply = (u[:, 3] < self.prob).float()
        return shift, torch.cat([b[:, None], c[:, None], h[:, None], order, apply[:, None]], dim=1).contiguous()


def draw_play_batch_augmentation(specs, B, T, device, generator=None):
    """batch["aug"] for a play batch: per camera a different draw for every frame of the window and for the goal frame,
    as the reference's dataset applies its transform to (T,3,H,W) windows and to the goal frame separately
    (play_dataset.py:405-412,246-247; RandomShiftsAug draws per leading index)."""
    aug = {"states": {}, "goal": {}, "pad": {}, "resize": {}}
    for cam, sp in specs.items():
        if sp.resize is not None:
            aug["resize"][cam] = sp.resize
        s, j = sp.draw(B * T, device, generator)
        aug["states"][cam] = {"shift": s.view(B, T, 2), "jitter": j.view(B, T, 8)}
        s, j = sp.draw(B, device, generator)
        aug["goal"][cam] = {"shift": s, "jitter": j}
        aug["pad"][cam] = sp.pad
    return aug
