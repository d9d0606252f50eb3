"""One process per GPU for the reference's UNCHANGED `train3dunet` / `predict3dunet`.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        -m pytorch3dunet_amd.launch train --config train_config.yml
    python -m pytorch3dunet_amd.launch predict --config test_config.yml          # 1 process; N processes shard the files

The reference's only parallelism is single-process `nn.DataParallel` (unet3d/trainer.py:202-205, predict.py:63-66).  Under plain
`torchrun` its `UNetTrainer` would checkpoint on every rank (trainer.py:382-403), write TensorBoard on every rank (:405-433) and
feed every rank the same patches (datasets/utils.py:399-422).  This launcher fixes exactly that, without editing a reference file:

  1. each rank sees ONE device (`HIP_VISIBLE_DEVICES` = its LOCAL_RANK's device, set before the HIP runtime starts), so the
     unchanged trainer neither wraps `nn.DataParallel` (trainer.py:203) nor rescales the batch (datasets/utils.py:399-403);
  2. the `sys.modules` seam of INTEGRATION.md: `pytorch3dunet.unet3d.{model,buildingblocks,se,predictor}` resolve to this
     package before the trainer / predictor import them;
  3. `create_trainer(config)` is the reference's own; the loaders it builds are re-wrapped with a per-rank `DistributedSampler`
     over the same `ConcatDataset` (same batch size, workers and collate function), staged through `DevicePrefetcher` on HIP;
  4. `parallel.attach(model)`: rank 0's parameters (after `resume` / `pre_trained`) are broadcast, the gradient all-reduce is
     overlapped with the encoder backward (RCCL over xGMI; gloo + hooks for `device: cpu`);
  5. the validation score is the SAMPLE-WEIGHTED mean over ranks (every rank validates its unpadded shard), i.e. the single-process
     score, so `ReduceLROnPlateau` and the best-score bookkeeping agree everywhere and with a one-process run; only rank 0 writes checkpoints, TensorBoard events and the config copy.
"""
from __future__ import annotations

import importlib
import os
import sys
from typing import Optional

_SEAM = ("model", "buildingblocks", "se", "predictor")


def rank_info():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def pin_device() -> Optional[str]:
    """Restrict this process to its LOCAL_RANK's device.  Must run before the HIP runtime initialises (i.e. before the first
    `torch.cuda` call); returns the device id string it selected, or None when there is nothing to do (single process)."""
    _, local_rank, world = rank_info()
    if world <= 1 and "LOCAL_RANK" not in os.environ:
        return None
    import torch

    if torch.cuda.is_initialized():
        raise RuntimeError("pytorch3dunet_amd.launch.pin_device() must run before the first torch.cuda call")
    listed = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES")
    if listed:
        ids = [v.strip() for v in listed.split(",") if v.strip() != ""]
        if local_rank >= len(ids):
            raise RuntimeError(f"LOCAL_RANK {local_rank} but only {len(ids)} visible device(s): {listed!r}")
        mine = ids[local_rank]
    else:
        mine = str(local_rank)
    # ONE variable: HIP applies HIP_VISIBLE_DEVICES and its CUDA_ alias as successive filters
    os.environ.pop("CUDA_VISIBLE_DEVICES", None)
    os.environ["HIP_VISIBLE_DEVICES"] = mine
    return mine


def install_seam() -> None:
    """`pytorch3dunet.unet3d.<name>` -> `pytorch3dunet_amd.unet3d.<name>` for the modules of the hot path, and forget reference
    modules that were imported before (they hold `from ... import get_model` bindings of the old modules)."""
    try:
        importlib.import_module("pytorch3dunet")
    except ImportError as e:  # pragma: no cover
        raise ImportError("pytorch3dunet_amd.launch drives the reference's own train / predict entry points: the "
                          "`pytorch3dunet` package (wolny/pytorch-3dunet) must be importable") from e
    for name in _SEAM:
        sys.modules[f"pytorch3dunet.unet3d.{name}"] = importlib.import_module(f"pytorch3dunet_amd.unet3d.{name}")
    for name in ("pytorch3dunet.unet3d.trainer", "pytorch3dunet.train", "pytorch3dunet.predict"):
        sys.modules.pop(name, None)
    import pytorch3dunet.unet3d as pkg

    for name in _SEAM:
        setattr(pkg, name, sys.modules[f"pytorch3dunet.unet3d.{name}"])
    # losses: the reference's OWN module stays (option handling, wrappers, every other loss); only the fused family — BCEDiceLoss,
    # DiceLoss, nn.BCEWithLogitsLoss: csrc/u3d_loss.hip — is patched into it
    from .unet3d.losses import install_fused

    install_fused(importlib.import_module("pytorch3dunet.unet3d.losses"))


def init_distributed(device: str) -> bool:
    """Process group over RCCL (`nccl`) for HIP devices, gloo for `device: cpu`.  False when this is a single process."""
    import torch
    import torch.distributed as dist

    rank, _, world = rank_info()
    if world <= 1 and "RANK" not in os.environ:
        return False
    if dist.is_initialized():
        return True
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if str(device) == "cuda":
        torch.cuda.set_device(0)  # the only visible one (pin_device)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    return True


class _NullWriter:
    """`SummaryWriter` of the ranks that do not log (trainer.py:176-178 builds one unconditionally)"""

    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        return lambda *a, **k: None


class _UnpaddedShard:
    """Sampler of the VALIDATION shard of one rank: indices rank, rank + world, ... of the (optionally shuffled) dataset order, NOT
    padded with duplicates — ranks may hold different numbers of samples (validation runs no collective per batch), and the
    sample-weighted reduction in `create_distributed_trainer` then reproduces the single-process average exactly."""

    def __init__(self, n: int, rank: int, world: int, shuffle: bool, seed: int):
        self.n, self.rank, self.world, self.shuffle, self.seed, self.epoch = n, rank, world, shuffle, seed, 0

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def __len__(self):
        return len(range(self.rank, self.n, self.world))

    def __iter__(self):
        import torch

        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            order = torch.randperm(self.n, generator=g).tolist()
        else:
            order = list(range(self.n))
        return iter(order[self.rank :: self.world])


class ShardedLoader:
    """A DataLoader over the SAME dataset, batch size, workers and collate function as the reference's, sharded per rank.
    Training: `DistributedSampler`, padded to equal length — every rank runs the same number of iterations, so the gradient
    collectives pair up; a new epoch (one pass of `UNetTrainer.train`, trainer.py:231) reshuffles with the epoch number as torch
    DDP recipes do.  Validation (`pad=False`): `_UnpaddedShard` — no duplicated samples, so the rank-weighted score equals the
    single-process one."""

    def __init__(self, loader, rank: int, world: int, seed: int = 0, pad: bool = True):
        from torch.utils.data import DataLoader, RandomSampler
        from torch.utils.data.distributed import DistributedSampler

        shuffle = isinstance(loader.sampler, RandomSampler)
        if pad or len(loader.dataset) < world:  # (fewer samples than ranks: an empty shard would have no score at all)
            self.sampler = DistributedSampler(loader.dataset, num_replicas=world, rank=rank, shuffle=shuffle, seed=seed, drop_last=False)
        else:
            self.sampler = _UnpaddedShard(len(loader.dataset), rank, world, shuffle, seed)
        kw = dict(batch_size=loader.batch_size, sampler=self.sampler, num_workers=loader.num_workers, collate_fn=loader.collate_fn,
                  pin_memory=loader.pin_memory, drop_last=loader.drop_last, timeout=loader.timeout,
                  worker_init_fn=loader.worker_init_fn)
        if loader.num_workers > 0:
            kw.update(multiprocessing_context=loader.multiprocessing_context, persistent_workers=loader.persistent_workers,
                      prefetch_factor=loader.prefetch_factor)
        self.loader = DataLoader(loader.dataset, **kw)
        self.dataset = loader.dataset
        self.batch_size = loader.batch_size
        self.epoch = 0

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        self.sampler.set_epoch(self.epoch)
        self.epoch += 1
        return iter(self.loader)


class _CountingLoader:
    """outermost wrapper of the validation loader: counts the samples of the batches the trainer actually CONSUMED in the current
    pass (`UNetTrainer._batch_size`, trainer.py:370-379: the first dimension of the input) — the weight of this rank's score"""

    def __init__(self, loader):
        self.loader = loader
        self.samples = 0

    def __len__(self):
        return len(self.loader)

    def __getattr__(self, name):
        return getattr(self.loader, name)

    def __iter__(self):
        self.samples = 0
        for batch in self.loader:
            first = batch[0] if isinstance(batch, (list, tuple)) else batch
            while isinstance(first, (list, tuple)):
                first = first[0]
            self.samples += int(first.shape[0])  # every batch handed out is scored before the trainer's `break` test (trainer.py:329-341)
            yield batch


def shard_loaders(loaders: dict, rank: int, world: int, seed: int = 0, device: str = "cpu", prefetch: bool = True) -> dict:
    out = {}
    for phase, loader in loaders.items():
        ld = ShardedLoader(loader, rank, world, seed, pad=(phase != "val")) if world > 1 else loader
        if prefetch and str(device) == "cuda":
            from .data import DevicePrefetcher

            ld = DevicePrefetcher(ld, "cuda")
        if phase == "val" and world > 1:
            ld = _CountingLoader(ld)
        out[phase] = ld
    return out


def create_distributed_trainer(config: dict, prefetch: bool = True):
    """The reference's `create_trainer(config)` (trainer.py:32-78) — model, loss, metric, loaders, optimizer, scheduler, resume —
    with the per-rank loaders, the gradient exchange and the rank-0-only side effects described in the module docstring."""
    import torch.distributed as dist

    from . import parallel

    T = importlib.import_module("pytorch3dunet.unet3d.trainer")
    rank, _, world = rank_info()
    distributed = dist.is_initialized()
    if not distributed:
        rank, world = 0, 1
    device = str(getattr(config.get("device"), "value", config.get("device")))
    seed = int(config.get("manual_seed") or 0)
    orig_loaders, orig_writer = T.get_train_loaders, T.SummaryWriter
    T.get_train_loaders = lambda cfg: shard_loaders(orig_loaders(cfg), rank, world, seed, device, prefetch)
    if rank != 0:
        T.SummaryWriter = _NullWriter
    try:
        trainer = T.create_trainer(config)
    finally:
        T.get_train_loaders, T.SummaryWriter = orig_loaders, orig_writer
    if distributed:
        import torch

        if isinstance(trainer.model, torch.nn.DataParallel):  # pragma: no cover  (pin_device prevents it)
            raise RuntimeError("several devices are visible to this rank: call launch.pin_device() before torch.cuda starts")
        trainer.grad_sync = parallel.attach(trainer.model, broadcast=True)
        if device == "cuda":
            parallel.cu_budget()  # U3D_RCCL_SLOTS=k: block slots the persistent convolution grids leave free for RCCL's kernels
        validate = trainer.validate

        def validate_all_ranks():
            # Sample-weighted mean over the WHOLE validation set: every rank's `val_scores.avg` (trainer.py:309-349, a
            # running average weighted by batch size) times the samples it consumed, summed over ranks, divided by the total — the
            # number a single process computes over the same data, so ReduceLROnPlateau and the best-checkpoint decision
            # (trainer.py:254-262) do not depend on the number of ranks.  (Image logging indices, `max_val_images`, stay per shard:
            # rank 0 logs images of ITS shard.)
            score = validate()
            val = trainer.loaders.get("val")
            n = val.samples if isinstance(val, _CountingLoader) else 1
            t = torch.tensor([float(score) * n, float(n)], dtype=torch.float64, device="cuda" if device == "cuda" else "cpu")
            dist.all_reduce(t)
            return (t[0] / t[1]).item()

        trainer.validate = validate_all_ranks
        if rank != 0:
            trainer._save_checkpoint = lambda is_best: None
    return trainer


def train_main(argv) -> None:
    """`train3dunet --config X` (pytorch3dunet/train.py:16-43), one process per GPU."""
    import random

    pin_device()
    install_seam()
    import torch
    from pytorch3dunet.unet3d.config import copy_config, load_config

    prefetch = True
    if "--no-prefetch" in argv:
        argv = [a for a in argv if a != "--no-prefetch"]
        prefetch = False
    sys.argv = ["train3dunet"] + list(argv)
    config, config_path = load_config()
    seed = config.get("manual_seed", None)
    if seed is not None:
        random.seed(seed)
        torch.manual_seed(seed)
    device = str(getattr(config["device"], "value", config["device"]))
    distributed = init_distributed(device)
    trainer = create_distributed_trainer(config, prefetch)
    rank = rank_info()[0] if distributed else 0
    if rank == 0:
        copy_config(config, config_path)
    trainer.fit()
    if distributed:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


def predict_main(argv) -> None:
    """`predict3dunet --config X` (pytorch3dunet/predict.py:43-88); with several processes the test FILES are dealt out
    round-robin (one loader per file, datasets/utils.py:426-470 — predictions of different files are independent)."""
    pin_device()
    install_seam()
    P = importlib.import_module("pytorch3dunet.predict")
    rank, _, world = rank_info()
    if world > 1:
        orig = P.get_test_loaders

        def mine(config):
            for i, loader in enumerate(orig(config)):
                if i % world == rank:
                    yield loader

        P.get_test_loaders = mine
    sys.argv = ["predict3dunet"] + list(argv)
    P.main()


def main(argv=None) -> None:
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] not in ("train", "predict"):
        raise SystemExit("usage: python -m pytorch3dunet_amd.launch {train|predict} --config <yaml> [reference CLI overrides]")
    (train_main if argv[0] == "train" else predict_main)(argv[1:])


if __name__ == "__main__":
    main()
