"""End-to-end epoch throughput of the three batch forms on one GPU: a WikiMEL-layout .npy set is
written to a scratch directory, then one training epoch at the reference's batch of 64 runs through
  gathered  - the reference's loader contract: host gather of 22 MB per mention, host-to-device copy, per-step token pooling
  indexed   - entity tables on the device, mention-side tensors from the host loader
  device    - the whole split on the device (DeviceSplit): no host work in the step
usage: python tools/epoch_bench.py [mentions] [entities] [scratch_dir]"""
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from drin_amd.config import wikimel_config  # noqa: E402
from drin_amd.data import (create_datasets, create_device_splits, create_indexed_datasets, load_entity_table,  # noqa: E402
                           write_synthetic_dataset)
from drin_amd.model import Model  # noqa: E402
from drin_amd.train import MELRunner, seed_everything  # noqa: E402

mentions = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
entities = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
root = sys.argv[3] if len(sys.argv) > 3 else tempfile.mkdtemp(prefix="drin_epoch_")
dev = torch.device("cuda")
cfg = wikimel_config().with_(batch_size=64, shuffle_train_data=True, num_epoch=1, test_epoch_interval=1)
t0 = time.perf_counter()
write_synthetic_dataset(cfg, root, sizes=(mentions, 64, 64), seed=3, num_entities=entities, lean=True)
print(f"wrote {mentions} mentions / {entities} entities to {root} in {time.perf_counter() - t0:.1f} s", flush=True)
N = cfg.num_candidates_model
for form in ("device", "indexed", "gathered"):
    seed_everything(cfg.seed)
    model = Model(cfg, precision="bf16x3").to(dev)
    table = None
    if form == "gathered":
        loaders = create_datasets(cfg, root, num_workers=8, mention_mmap="r", entity_mmap="r")
    else:
        table = load_entity_table(cfg, root, dev, entity_mmap="r")
        loaders = (create_device_splits(cfg, root, dev) if form == "device"
                   else create_indexed_datasets(cfg, root, num_workers=8, mention_mmap="r"))
    runner = MELRunner(cfg, model, dev, entity_table=table)
    opt = torch.optim.Adam(model.parameters(), lr=cfg.learning_rate)
    runner.run_epoch(loaders[1], 1, opt)                   # warm-up: one training pass over the valid split
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    log = runner.run_epoch(loaders[0], 0, opt)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = len(loaders[0])
    print(f"{form:9s}: {steps} steps in {dt:7.3f} s = {dt / steps * 1e3:8.2f} ms/step, {mentions / dt:9.1f} mentions/s, "
          f"{mentions * N / dt / 1e6:6.3f} M pairs/s, train loss {log.loss:.5f}", flush=True)
    del model, runner, loaders, table
    torch.cuda.empty_cache()
if len(sys.argv) <= 3:
    shutil.rmtree(root, ignore_errors=True)
