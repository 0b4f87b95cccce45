"""GPU: the whole data-parallel train step with two ranks sharing the one GPU of the test box (gloo transport, since
RCCL refuses two ranks on one device): gradient buckets are announced from inside the backward (autograd thread,
side stream), averaged, and both ranks must end the step with identical weights that differ from an un-exchanged
run.  The RCCL path differs only in the backend string."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
V, E, L = 60, 64, 8


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _model():
    from acvae_amd.decoder import VAERNNBahdanauAttnDecoder
    from acvae_amd.encoder import Cnn10
    from acvae_amd.vae_model import Hybrid_VAEModel
    torch.manual_seed(5)
    dec = VAERNNBahdanauAttnDecoder(vocab_size=V, enc_mem_size=E, embed_size=E, hidden_size=E, attn_size=E)
    return Hybrid_VAEModel(Cnn10(64, 512), dec, posterior_model="PosteriorRNN_hybrid", posterior_args={"hidden_size": E},
                           prior_model="PriorRNN", prior_args={"hidden_size": E}).cuda().train()


def _batch(seed):
    g = torch.Generator().manual_seed(seed)
    feats = torch.randn(3, 64, 64, generator=g)
    caps = torch.zeros(3, L); caps[:, 0] = 1; caps[:, -1] = 2
    caps[:, 1:-1] = torch.randint(4, V, (3, L - 2), generator=g).float()
    return feats, caps, np.full(3, 64), np.full(3, L)


def _worker(rank, world, port, q):
    import random
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from acvae_amd.trainer import TrainStep
    model = _model()
    model.encoder.p_block = model.encoder.p_fc = 0.0
    ts = TrainStep(model, V)
    announced, ready = [], ts.exchange.ready
    ts.exchange.ready = lambda i: (announced.append(i), ready(i))[1]
    feats, caps, fl, cl = _batch(10 + rank)             # each rank its own shard
    torch.manual_seed(1); random.seed(1)
    parts = ts.step(feats.cuda(), fl.copy(), caps, cl, 1.0, 0, 0.5)
    torch.cuda.synchronize()
    q.put((rank, ts.flat_p.cpu().numpy().copy(), float(parts["loss"]), announced))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_train_step_on_one_gpu():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = dict((r, (w, l, a)) for r, w, l, a in (q.get(timeout=300) for _ in range(world)))
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert np.array_equal(res[0][0], res[1][0]), "ranks diverged after the averaged step"
    assert res[0][1] != res[1][1]                        # different shards -> different local losses
    for r in range(world):                               # buckets: 0 text, 1 encoder blocks 1-3 (+bn0), 2 last ConvBlock
        seq = res[r][2]
        assert set(seq) == {0, 1, 2}, seq
        assert seq.index(2) < seq.index(1), seq          # the deepest block's gradients go on the wire before the rest
    # single-process run of rank 0's shard without exchange must differ (the exchange really happened)
    import random
    from acvae_amd.trainer import TrainStep
    model = _model()
    model.encoder.p_block = model.encoder.p_fc = 0.0
    ts = TrainStep(model, V)
    feats, caps, fl, cl = _batch(10)
    torch.manual_seed(1); random.seed(1)
    ts.step(feats.cuda(), fl.copy(), caps, cl, 1.0, 0, 0.5)
    torch.cuda.synchronize()
    assert not np.array_equal(ts.flat_p.cpu().numpy(), res[0][0])
