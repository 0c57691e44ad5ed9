"""`.npy` -> 15-tuple batches, the boundary feeder of the path (`drin/data.py`).

File names, array layouts, the WikiDiverse reshapes (`data.py:30-38`), the WikiMEL entity-table
gather through `qid2idx.json` (`data.py:40-46,87-93`), the +1 CLS shift of the span (`data.py:113-114`)
and the one-hot answer table with its all-zero last row (`data.py:159-161`) follow the reference, so a
directory preprocessed by the reference's `preprocess/*.py` loads unchanged.  Since neither dataset is
available offline, `write_synthetic_dataset` produces a directory of the same layout from seeds.
"""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from . import synth
from .config import DrinConfig
from .npy_stream import NpyWriter, open_npy

SPLITS = ("train", "valid", "test")


def _load(path: str, mmap: Optional[str] = None) -> np.ndarray:
    if mmap == "r":
        return open_npy(path)          # header checked against the file size first (unclosed / truncated writers)
    return np.load(path, mmap_mode=mmap)


class MELData(Dataset):
    """One split.  `__getitem__` returns the 15-tuple of `drin/data.py:110-126`."""

    def __init__(self, cfg: DrinConfig, root: str, split: str, shared: Dict[str, np.ndarray],
                 mention_mmap: Optional[str] = None):
        self.cfg = cfg
        N, D, R = cfg.num_candidates_model, cfg.bert_embed_dim, cfg.resnet_embed_dim
        p = lambda name: os.path.join(root, name)  # noqa: E731
        self.onehot = shared["onehot"]
        if cfg.dataset_name == "wikidiverse":
            self.entity_text_feature = shared[f"entity_text_{split}"].reshape((-1, N, D))            # data.py:31
            self.entity_image_feature = shared[f"entity_image_{split}"].reshape((-1, N, R))          # data.py:32
            self.entity_object_feature = shared[f"entity_object_{split}"].reshape((-1, N, cfg.object_topk_entity, R))
            self.entity_object_score = shared[f"entity_object_score_{split}"].reshape((-1, N, cfg.object_topk_entity))
        else:
            self.entity_text_feature = shared["entity_text"]
            self.entity_text_mask = shared["entity_text_mask"]
            self.entity_image_feature = shared["entity_image"]
            self.entity_object_feature = shared["entity_object"]
            self.entity_object_score = shared["entity_object_score"]
            with open(p("qid2idx.json")) as f:
                self.qid2idx = json.load(f)                                                         # data.py:42-43
            self.entity_qid = _load(p(f"entity-name-raw_{split}.npy")).reshape((-1, N))              # data.py:45-46
        self.mention_text_feature = _load(p(f"mention-text-feature_{split}.npy"), mention_mmap)
        self.mention_text_mask = _load(p(f"mention-text-mask_{split}.npy"))
        self.mention_start_pos = _load(p(f"start-pos_{split}.npy"))
        self.mention_end_pos = _load(p(f"end-pos_{split}.npy"))
        self.mention_image_feature = _load(p(f"mention-image-feature_{split}.npy"), mention_mmap)
        self.mention_object_feature = _load(p(f"mention-object-feature_{split}.npy"), mention_mmap)
        self.mention_object_score = _load(p(f"mention-object-score_{split}.npy"))
        self.miet_similarity = _load(p(f"similarity-miet_{split}.npy"))
        self.mtei_similarity = _load(p(f"similarity-eimt_{split}.npy"))
        self.answer = _load(p(f"answer_{split}.npy"))
        n = len(self.answer)
        for name in ("mention_text_feature", "mention_start_pos", "mention_image_feature", "mention_object_feature",
                     "miet_similarity"):
            if len(getattr(self, name)) != n:                                                       # data.py:73-80
                raise ValueError(f"{name} has {len(getattr(self, name))} rows, answer has {n}")

    def __len__(self) -> int:
        return len(self.answer)

    @staticmethod
    def _t(x) -> torch.Tensor:
        return torch.as_tensor(np.array(x) if isinstance(x, np.memmap) else x)

    def __getitem__(self, idx):
        t = self._t
        entity_text_mask = 0                                                                        # data.py:86
        if self.cfg.dataset_name == "wikimel":
            rows = [self.qid2idx[str(q)] for q in self.entity_qid[idx]]                              # data.py:88
            etf, emask = t(self.entity_text_feature[rows]), t(self.entity_text_mask[rows])
            eimg, eobj, escore = (t(self.entity_image_feature[rows]), t(self.entity_object_feature[rows]),
                                  t(self.entity_object_score[rows]))
            entity_text_mask = emask
        else:
            etf, eimg = t(self.entity_text_feature[idx]), t(self.entity_image_feature[idx])
            eobj, escore = t(self.entity_object_feature[idx]), t(self.entity_object_score[idx])
        return (
            t(self.mention_text_feature[idx]), t(self.mention_text_mask[idx]),
            t(self.mention_start_pos[idx]) + 1, t(self.mention_end_pos[idx]) + 1,                    # data.py:113-114
            t(self.mention_image_feature[idx]), t(self.mention_object_feature[idx]), t(self.mention_object_score[idx]),
            etf, entity_text_mask, eimg, eobj, escore,
            t(self.miet_similarity[idx]), t(self.mtei_similarity[idx]),
            t(self.onehot[self.answer[idx]]),                                                       # data.py:109
        )


def create_datasets(cfg: DrinConfig, root: str, batch_size: Optional[int] = None, num_workers: int = 0,
                    rank: int = 0, world_size: int = 1, mention_mmap: Optional[str] = None,
                    entity_mmap: Optional[str] = None) -> List[DataLoader]:
    """`create_datasets()` of `drin/data.py:158-200`: [train, valid, test] loaders.

    With `world_size > 1` every rank iterates its own strided shard `perm[rank::world]` of each split (mentions
    are independent); shuffling of the train split uses the same seeded permutation on all ranks so
    the shards stay disjoint.  The TRAIN shards are padded with wrapped-around mentions to one common length
    (like `DistributedSampler`): every rank then runs the same number of steps with the same batch sizes, which the
    per-step gradient all-reduce (and the gathered global-batch loss) need.  Evaluation shards are not padded - every
    mention counts exactly once in the metrics - and evaluation makes no per-step collective.
    """
    N = cfg.num_candidates_model
    onehot = np.concatenate([np.eye(N - 1, dtype=np.uint8), np.zeros((1, N - 1), dtype=np.uint8)], 0)   # data.py:159-161
    shared: Dict[str, np.ndarray] = {"onehot": onehot}
    p = lambda name: os.path.join(root, name)  # noqa: E731
    if cfg.dataset_name == "wikimel":                                                               # data.py:163-175
        shared["entity_text"] = _load(p("entity-attr-feature.npy"), entity_mmap)
        shared["entity_text_mask"] = _load(p("entity-attr-mask.npy"))
        shared["entity_image"] = _load(p("entity-image-feature_all.npy"), entity_mmap)
        shared["entity_object"] = _load(p("entity-object-feature_all.npy"), entity_mmap)
        shared["entity_object_score"] = _load(p("entity-object-score_all.npy"))
    else:                                                                                           # data.py:188-200
        for s in SPLITS:
            shared[f"entity_text_{s}"] = _load(p(f"entity-attr-feature_{s}.npy"), entity_mmap)
            shared[f"entity_image_{s}"] = _load(p(f"entity-image-feature_{s}.npy"), entity_mmap)
            shared[f"entity_object_{s}"] = _load(p(f"entity-object-feature_{s}.npy"), entity_mmap)
            shared[f"entity_object_score_{s}"] = _load(p(f"entity-object-score_{s}.npy"))
    loaders = []
    for s in SPLITS:
        ds = MELData(cfg, root, s, shared, mention_mmap)
        shuffle = s == "train" and cfg.shuffle_train_data                                           # data.py:155
        sampler = ShardSampler(len(ds), rank, world_size, shuffle, cfg.seed, pad=s == "train") if (world_size > 1 or shuffle) else None
        loaders.append(DataLoader(ds, batch_size or cfg.batch_size, shuffle=False, sampler=sampler, num_workers=num_workers))
    return loaders


class IndexedMELData(MELData):
    """WikiMEL split in table form: `__getitem__` returns the seven mention-side tensors,
    the candidate rows of the entity tables `[N]` int64 (what `qid2idx` maps the QIDs to, data.py:88), the two
    similarity rows and the answer - 11 items, ~1 MB per mention instead of 22 MB.  The tables themselves go to
    the device once (`load_entity_table`) and are gathered inside the HIP stream kernel."""

    def __getitem__(self, idx):
        t = self._t
        rows = np.asarray([self.qid2idx[str(q)] for q in self.entity_qid[idx]], dtype=np.int64)
        return (
            t(self.mention_text_feature[idx]), t(self.mention_text_mask[idx]),
            t(self.mention_start_pos[idx]) + 1, t(self.mention_end_pos[idx]) + 1,
            t(self.mention_image_feature[idx]), t(self.mention_object_feature[idx]), t(self.mention_object_score[idx]),
            torch.from_numpy(rows), t(self.miet_similarity[idx]), t(self.mtei_similarity[idx]),
            t(self.onehot[self.answer[idx]]),
        )


def load_entity_table(cfg: DrinConfig, root: str, device="cpu", entity_mmap: Optional[str] = None):
    """The five WikiMEL entity tables of `drin/data.py:163-175` as a `drin_amd.model.EntityTable` on `device`."""
    from .model import EntityTable

    p = lambda name: os.path.join(root, name)  # noqa: E731
    up = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(device)  # noqa: E731
    return EntityTable(up(_load(p("entity-attr-feature.npy"), entity_mmap)), up(_load(p("entity-attr-mask.npy"))),
                       up(_load(p("entity-image-feature_all.npy"), entity_mmap)),
                       up(_load(p("entity-object-feature_all.npy"), entity_mmap)),
                       up(_load(p("entity-object-score_all.npy"))))


def create_indexed_datasets(cfg: DrinConfig, root: str, batch_size: Optional[int] = None, num_workers: int = 0,
                            rank: int = 0, world_size: int = 1, mention_mmap: Optional[str] = None) -> List[DataLoader]:
    """[train, valid, test] loaders of `IndexedMELData` (WikiMEL only)."""
    if cfg.dataset_name != "wikimel":
        raise ValueError("the table form exists for the wikimel layout only (drin/data.py:40-46)")
    N = cfg.num_candidates_model
    onehot = np.concatenate([np.eye(N - 1, dtype=np.uint8), np.zeros((1, N - 1), dtype=np.uint8)], 0)
    empty = np.zeros((0,), dtype=np.float32)
    shared = {"onehot": onehot, "entity_text": empty, "entity_text_mask": empty, "entity_image": empty,
              "entity_object": empty, "entity_object_score": empty}
    loaders = []
    for s in SPLITS:
        ds = IndexedMELData(cfg, root, s, shared, mention_mmap)
        shuffle = s == "train" and cfg.shuffle_train_data
        sampler = ShardSampler(len(ds), rank, world_size, shuffle, cfg.seed, pad=s == "train") if (world_size > 1 or shuffle) else None
        loaders.append(DataLoader(ds, batch_size or cfg.batch_size, shuffle=False, sampler=sampler, num_workers=num_workers))
    return loaders


class DeviceSplit:
    """One split resident on the device (WikiMEL in table form; WikiDiverse as its per-mention tensors) (".npy -> HBM once"): the seven mention-side tensors,
    the candidate rows of the entity tables, the two similarity matrices and the answers of EVERY mention of the split
    are uploaded once (WikiMEL train: 18 k mentions x 0.82 MB = 15 GB of the 288 GB); iterating yields the same
    11-item batches as a `DataLoader` over `IndexedMELData` - same order, same sharding, same shuffling - but as slices /
    index-selects of device tensors: no worker processes, no host gather, no host-to-device copy in the step."""

    def __init__(self, ds: MELData, device, batch_size: int, sampler: Optional["ShardSampler"]):
        dev = torch.device(device)
        up = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        mention = [up(ds.mention_text_feature), up(ds.mention_text_mask),
                   up(ds.mention_start_pos) + 1, up(ds.mention_end_pos) + 1,                         # data.py:113-114
                   up(ds.mention_image_feature), up(ds.mention_object_feature), up(ds.mention_object_score)]
        if isinstance(ds, IndexedMELData):
            rows = np.asarray([[ds.qid2idx[str(q)] for q in qs] for qs in ds.entity_qid], dtype=np.int64)   # data.py:88
            self.tensors = mention + [up(rows), up(ds.miet_similarity), up(ds.mtei_similarity)]
        elif ds.cfg.dataset_name == "wikidiverse":
            # WikiDiverse stores per-mention candidate tensors (data.py:31-34): the whole 15-tuple lives on the device
            n = len(ds)
            self.tensors = mention + [up(ds.entity_text_feature), torch.zeros(n, dtype=torch.int64, device=dev),   # data.py:86
                                      up(ds.entity_image_feature), up(ds.entity_object_feature), up(ds.entity_object_score),
                                      up(ds.miet_similarity), up(ds.mtei_similarity)]
        else:
            raise ValueError("a WikiMEL split goes on the device in table form: build it from IndexedMELData")
        self.onehot = up(ds.onehot)
        self.answer = up(np.asarray(ds.answer).astype(np.int64))
        self.n, self.batch_size, self.sampler, self.device = len(ds), batch_size, sampler, dev

    def __len__(self) -> int:
        n = len(self.sampler) if self.sampler is not None else self.n
        return (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        if self.sampler is None:
            for b0 in range(0, self.n, self.batch_size):                  # contiguous: views, nothing is copied
                sl = slice(b0, min(self.n, b0 + self.batch_size))
                yield [t[sl] for t in self.tensors] + [self.onehot[self.answer[sl]]]
        else:
            # one permuted copy of the rank's shard per epoch (a few ms for 15 GB), then views: the step itself launches
            # nothing for its batch - at the reference's batch of 64 the training step is close to host-bound
            order = torch.as_tensor(list(self.sampler), dtype=torch.int64).to(self.device)
            shard = [t.index_select(0, order) for t in self.tensors] + [self.onehot[self.answer.index_select(0, order)]]
            for b0 in range(0, order.numel(), self.batch_size):
                yield [t[b0:b0 + self.batch_size] for t in shard]


def create_device_splits(cfg: DrinConfig, root: str, device, batch_size: Optional[int] = None, rank: int = 0,
                         world_size: int = 1, mention_mmap: Optional[str] = None) -> List[DeviceSplit]:
    """[train, valid, test] `DeviceSplit`s: drop-in for `create_indexed_datasets` (WikiMEL: table form, pass the
    `EntityTable` to `MELRunner`) or `create_datasets` (WikiDiverse: the 15-tuples themselves) in `MELRunner.fit`."""
    if cfg.dataset_name == "wikimel":
        loaders = create_indexed_datasets(cfg, root, batch_size, 0, rank, world_size, mention_mmap)
    else:
        loaders = create_datasets(cfg, root, batch_size, 0, rank, world_size, mention_mmap, mention_mmap)
    return [DeviceSplit(ld.dataset, device, batch_size or cfg.batch_size, ld.sampler if isinstance(ld.sampler, ShardSampler) else None)
            for ld in loaders]


class ShardSampler(torch.utils.data.Sampler):
    """Rank r of w draws indices `order[r::w]` of one permutation shared by all ranks (re-seeded per epoch).
    `pad`: `order` is first extended with its own head to a multiple of w (as `DistributedSampler` does), so that all
    ranks draw the same number of indices - required wherever ranks meet in a collective every step."""

    def __init__(self, n: int, rank: int, world: int, shuffle: bool, seed: int, pad: bool = False):
        self.n, self.rank, self.world, self.shuffle, self.seed, self.epoch = n, rank, world, shuffle, seed, 0
        self.pad = pad and world > 1

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def _order(self):
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            order = torch.randperm(self.n, generator=g).tolist()
        else:
            order = list(range(self.n))
        if self.pad and self.n % self.world:
            extra = self.world - self.n % self.world
            order += (order * (extra // max(self.n, 1) + 1))[:extra]
        return order

    def __iter__(self):
        return iter(self._order()[self.rank::self.world])

    def __len__(self) -> int:
        if self.pad:
            return (self.n + self.world - 1) // self.world if self.n else 0
        return len(range(self.rank, self.n, self.world))


def write_synthetic_dataset(cfg: DrinConfig, root: str, sizes=(256, 64, 64), seed: int = 1, num_entities: int = 512,
                            learnable: float = 0.0, lean: bool = False) -> None:
    """A directory in the reference's preprocessed layout, drawn from `synth` seeds.

    `learnable` > 0 (WikiDiverse layout only) plants a signal: the gold candidate's text feature becomes the mention's
    span mean plus `1 / learnable` times its original noise, so that a model can learn to rank it first - a stand-in
    task for end-to-end training checks (the real datasets are not available offline).

    WikiDiverse stores per-split candidate tensors flattened over (mention, candidate); WikiMEL stores
    one entity table plus per-split QID lists and `qid2idx.json` (`preprocess/bert.py:100-109`,
    `preprocess/resnet.py:159-185`, `preprocess/clip.py:143`).  Span positions are stored WITHOUT the CLS
    shift (the loader adds it); answers are indices into the one-hot table.
    """
    os.makedirs(root, exist_ok=True)
    N = cfg.num_candidates_model

    def save(name, arr):
        arr = np.asarray(arr)
        if arr.dtype.kind not in "iuf":                      # QID strings: not a streaming-writer type (utils.py:148-166)
            np.save(os.path.join(root, name), arr)
            return
        with NpyWriter(os.path.join(root, name)) as w:       # item by item, as preprocess/*.py write them
            w.extend(arr if arr.ndim > 1 else (np.asarray(v) for v in arr))   # 0-d items for 1-D files

    wm = cfg.dataset_name == "wikimel"
    if wm:
        tab = synth.make_batch(cfg.with_(num_candidates_data=num_entities - 1), 1, seed + 1000, as_torch=False)
        save("entity-attr-feature.npy", tab[7][0])                   # [E, T, D]
        save("entity-attr-mask.npy", tab[8][0])                      # [E, T]
        save("entity-image-feature_all.npy", tab[9][0])              # [E, 1, R]
        save("entity-object-feature_all.npy", tab[10][0])            # [E, Ke, 1, R]
        save("entity-object-score_all.npy", tab[11][0])              # [E, Ke]
        qids = [f"Q{100000 + 7 * i}" for i in range(num_entities)]
        with open(os.path.join(root, "qid2idx.json"), "w") as f:
            json.dump({q: i for i, q in enumerate(qids)}, f)
    for split, m in zip(SPLITS, sizes):
        # (`lean`, WikiMEL layout: the per-pair entity tensors of the draw are never stored - only QIDs are - so draw them
        #  with one token per entity: 0.3 MB instead of 20 MB of host memory per mention; a different random stream)
        draw_cfg = cfg.with_(max_entity_attr_token_len=1) if (lean and wm) else cfg
        b = synth.make_batch(draw_cfg, m, seed + SPLITS.index(split), as_torch=False)
        if learnable > 0 and not wm:
            gold = np.where(b[14].any(1), b[14].argmax(1), -1)
            for i in np.nonzero(gold >= 0)[0]:
                span = b[0][i, b[2][i]: b[3][i]].mean(0)
                b[7][i, gold[i]] = span + b[7][i, gold[i]] / learnable
        save(f"mention-text-feature_{split}.npy", b[0])
        save(f"mention-text-mask_{split}.npy", b[1])
        save(f"start-pos_{split}.npy", b[2] - 1)
        save(f"end-pos_{split}.npy", b[3] - 1)
        save(f"mention-image-feature_{split}.npy", b[4])
        save(f"mention-object-feature_{split}.npy", b[5])
        save(f"mention-object-score_{split}.npy", b[6])
        save(f"similarity-miet_{split}.npy", b[12])
        save(f"similarity-eimt_{split}.npy", b[13])
        save(f"answer_{split}.npy", np.where(b[14].any(1), b[14].argmax(1), N - 1).astype(np.int64))
        if wm:
            g = np.random.Generator(np.random.Philox(key=[seed, 50 + SPLITS.index(split)]))
            pick = g.integers(0, num_entities, size=(m, N))
            save(f"entity-name-raw_{split}.npy", np.array(qids)[pick].reshape(-1))
        else:
            save(f"entity-attr-feature_{split}.npy", b[7].reshape(m * N, -1))
            save(f"entity-image-feature_{split}.npy", b[9].reshape(m * N, -1))
            save(f"entity-object-feature_{split}.npy", b[10].reshape(m * N, cfg.object_topk_entity, -1))
            save(f"entity-object-score_{split}.npy", b[11].reshape(m * N, cfg.object_topk_entity))
