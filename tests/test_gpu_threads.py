"""-m gpu: `include/drin_hip.h` promises re-entrant entry points (no mutable global state, every launch on the caller's
stream).  Two host threads, each with its own stream, model and batch, score and train at the same time - ctypes drops the GIL
inside every library call, so the library's host code really runs concurrently - and every result must equal, bit for bit, what
the same model computes alone."""
import threading

import pytest
import torch

from drin_amd import synth
from drin_amd.config import DrinConfig, wikimel_config
from drin_amd.metrics import TripletLoss
from drin_amd.model import EntityTable, IndexedBatch, Model

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _work(cfg, seed, B, table_form):
    sd = synth.make_state_dict(cfg, seed)
    batch = synth.make_device_batch(cfg, B, seed + 1, DEV)
    ib = None
    if table_form:
        E = 97
        tab = synth.make_batch(cfg.with_(num_candidates_data=E - 1), 1, seed + 2)
        table = EntityTable(tab[7][0], tab[8][0] if cfg.token_level_entities else None, tab[9][0], tab[10][0], tab[11][0]).to(DEV)
        table.enable_cache(True, format="mixed_f16" if seed % 2 else "f32")
        cand = torch.randint(0, E, (B, cfg.num_candidates_model), generator=torch.Generator().manual_seed(seed)).to(DEV)
        ib = IndexedBatch(batch[:7], table, cand, batch[12], batch[13])

    def run():
        """inference (folded path), table form through the per-entity cache, one training step's gradients"""
        model = Model(cfg).to(DEV)
        model.load_state_dict(sd)
        model.eval()
        with torch.no_grad():
            out = [model(batch[:14]).clone()]
            if ib is not None:
                out.append(model(ib).clone())
        model.train()
        loss = TripletLoss(cfg.triplet_margin)(batch[14], model(batch[:14]))
        loss.backward()
        out.append(loss.detach().clone())
        out += [p.grad.clone() for p in model.parameters() if p.grad is not None]
        return out
    return run


def test_two_host_threads_on_two_streams_get_the_bits_of_a_lone_caller():
    jobs = [_work(wikimel_config(max_entity_attr_token_len=5, num_candidates_data=40), 11, 48, False),
            _work(DrinConfig(num_candidates_data=20), 20, 300, True)]
    alone = [job() for job in jobs]
    torch.cuda.synchronize()
    results, errors = [None, None], []

    def worker(i):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                for _ in range(6):
                    results[i] = jobs[i]()
                stream.synchronize()
        except BaseException as e:   # noqa: BLE001 - reported by the main thread
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in threads), "a worker thread hung"
    assert not errors, errors
    torch.cuda.synchronize()
    for i in range(2):
        assert len(results[i]) == len(alone[i])
        for got, ref in zip(results[i], alone[i]):
            assert torch.equal(torch.nan_to_num(got), torch.nan_to_num(ref)), f"thread {i}: a result differs from the lone caller's"
