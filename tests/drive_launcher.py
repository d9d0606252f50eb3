"""TEST INFRASTRUCTURE — drive `python -m pytorch3dunet_amd.launch train --config <yaml>` against the reference's UNMODIFIED
train entry point (/root/reference/pytorch3dunet/train.py:16-43 → trainer.py:32-78 create_trainer → UNetTrainer.fit), with a real
YAML file, the reference's StandardHDF5Dataset / SliceBuilder / transforms / get_train_loaders / MeanIoU / create_optimizer.

    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P tests/drive_launcher.py <workdir> dist
    python tests/drive_launcher.py <workdir> single

Stand-ins (no numerics): permissive skimage / tensorboard modules (oracle/ref_import.import_reference_runtime) and the in-memory
h5py of tests/fake_h5py.py holding the same seeded volumes in every process.  `single` runs ONE process that, at every iteration,
trains on the concatenation of what the two ranks of `dist` see (same DistributedSampler arithmetic), which is what data
parallelism must reproduce for a per-sample-mean loss (BCEWithLogitsLoss; GroupNorm statistics are per sample).
Writes <workdir>/params_<mode>_rank<r>.pt and prints one RESULT JSON line per process.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "pytorch-3dunet_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

CONFIG = """
device: cpu
manual_seed: 7
model:
  name: UNet3D
  in_channels: 1
  out_channels: 1
  f_maps: [8, 16]
  num_groups: 4
  layer_order: gcr
  final_sigmoid: true
  is_segmentation: true
loss:
  name: BCEWithLogitsLoss
optimizer:
  name: SGD
  learning_rate: 0.05
  momentum: 0.9
eval_metric:
  name: MeanIoU
trainer:
  checkpoint_dir: {ckpt}
  max_num_epochs: 2
  max_num_iterations: 3
  validate_after_iters: 2
  log_after_iters: 1
  eval_score_higher_is_better: true
loaders:
  dataset: StandardHDF5Dataset
  batch_size: 1
  num_workers: 0
  raw_internal_path: raw
  label_internal_path: label
  train:
    file_paths: [mem_train_a.h5, mem_train_b.h5]
    slice_builder: {{name: SliceBuilder, patch_shape: [4, 64, 64], stride_shape: [4, 64, 64]}}
    transformer:
      raw: [{{name: Standardize}}, {{name: ToTensor, expand_dims: true}}]
      label: [{{name: ToTensor, expand_dims: true}}]
  val:
    file_paths: [mem_val.h5]
    slice_builder: {{name: SliceBuilder, patch_shape: [4, 64, 64], stride_shape: [4, 64, 64]}}
    transformer:
      raw: [{{name: Standardize}}, {{name: ToTensor, expand_dims: true}}]
      label: [{{name: ToTensor, expand_dims: true}}]
"""


def make_files():
    import numpy as np

    import fake_h5py

    fake_h5py.install()
    rs = np.random.RandomState(3)
    for name, shape in (("mem_train_a.h5", (4, 64, 128)), ("mem_train_b.h5", (8, 64, 64)), ("mem_val.h5", (4, 64, 128))):
        with fake_h5py.File(name, "w") as f:
            raw = rs.randn(*shape).astype("float32")
            f.create_dataset("raw", data=raw)
            f.create_dataset("label", data=(raw > 0.2).astype("float32"))


class _GlobalBatches:
    """what `world` ranks see at each iteration, concatenated: the single-process twin of launch.ShardedLoader"""

    def __init__(self, loader, world, seed):
        from pytorch3dunet_amd.launch import ShardedLoader

        self.shards = [ShardedLoader(loader, r, world, seed) for r in range(world)]

    def __len__(self):
        return len(self.shards[0])

    def __iter__(self):
        import torch

        for parts in zip(*[iter(s) for s in self.shards]):
            yield tuple(torch.cat([p[i] for p in parts], dim=0) for i in range(len(parts[0])))


def main():
    workdir, mode = sys.argv[1], sys.argv[2]
    from ref_import import import_reference_runtime

    import_reference_runtime()
    make_files()
    import torch

    from pytorch3dunet_amd import launch

    torch.set_num_threads(2)
    rank = int(os.environ.get("RANK", "0"))
    ckpt = os.path.join(workdir, f"ckpt_{mode}")
    cfg_path = os.path.join(workdir, f"cfg_{mode}_{rank}.yml")
    with open(cfg_path, "w") as fh:
        fh.write(CONFIG.format(ckpt=ckpt))
    os.makedirs(os.path.join(ckpt, "logs"), exist_ok=True)  # (the real SummaryWriter creates it; copy_config scans it, config.py:101-113)
    events = {"saves": 0, "writers": 0}

    if mode == "dist":
        # the launcher end to end: python -m pytorch3dunet_amd.launch train --config <yaml>
        import pytorch3dunet_amd.launch as L

        orig_create = L.create_distributed_trainer

        def spy(config, prefetch=True):
            tr = orig_create(config, prefetch)
            save = tr._save_checkpoint

            def counted(is_best):
                events["saves"] += 1 if save.__name__ != "<lambda>" else 0
                return save(is_best)

            tr._save_checkpoint = counted
            events["trainer"] = tr
            events["writer_type"] = type(tr.writer).__name__
            return tr

        L.create_distributed_trainer = spy
        L.main(["train", "--config", cfg_path])
        tr = events["trainer"]
    else:
        launch.install_seam()
        from pytorch3dunet.unet3d.config import TorchDevice
        import yaml

        config = yaml.safe_load(open(cfg_path))
        config["device"] = TorchDevice(config["device"])
        torch.manual_seed(config["manual_seed"])
        tr = launch.create_distributed_trainer(config)
        tr.loaders = {k: _GlobalBatches(v, 2, config["manual_seed"]) for k, v in tr.loaders.items()}
        events["writer_type"] = type(tr.writer).__name__
        tr.fit()
    sd = {k: v.detach().clone() for k, v in tr.model.state_dict().items()}
    torch.save(sd, os.path.join(workdir, f"params_{mode}_rank{rank}.pt"))
    files = sorted(os.listdir(ckpt)) if os.path.isdir(ckpt) else []
    print("RESULT " + json.dumps({"mode": mode, "rank": rank, "iterations": tr.num_iterations, "writer": events["writer_type"],
                                  "saves": events["saves"], "ckpt_files": files, "best": float(tr.best_eval_score),
                                  "train_batches": len(tr.loaders["train"]),
                                  "sync": type(getattr(tr, "grad_sync", None)).__name__,
                                  "launched": getattr(getattr(tr, "grad_sync", None), "launched", 0)}), flush=True)


if __name__ == "__main__":
    main()
