"""Replay data path: play-window / goal sampling, the uint8 dataset resident in HBM (or in pinned
host memory behind a copy stream), GPU augmentations."""
