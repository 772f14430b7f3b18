"""world_size-2 data parallelism on CPU (gloo): the collective helpers of dppo_amd.parallel and the equivalence
claim behind them -- per-rank losses normalised with the POOLED minibatch moments, SUM-reduced, equal the
single-process loss and gradients.  The per-rank compute here is the CPU oracle (this is a test); on the GPU the
same contract is implemented by dppo_ppo_loss_fwd_bwd(global_moments=...)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import dppo_oracle as O
from tests.test_oracle_golden import make_cfg

KW = dict(denoising_steps=20, ft_denoising_steps=10, clip_ploss_coef=0.1, clip_ploss_coef_base=0.01,
          gamma_denoising=0.99, randn_clip_value=3)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def make_problem():
    a, c = O.named_specs("hopper")
    cfg = make_cfg(a, KW)
    rs = np.random.RandomState(11)
    R, Kft = 24, 10
    state = torch.from_numpy(rs.uniform(-1, 1, size=(R, 1, 11)).astype(np.float32))
    noise = torch.from_numpy(rs.randn(21, R, 4, 3).astype(np.float32))
    base, ft, cr = O.init_params(a, 61), O.init_params(a, 62), O.init_params(c, 63)
    _, chains = O.sample_chain(cfg, a, base, ft, state, noise)
    with torch.no_grad():
        logp = O.chain_logprob(cfg, a, base, ft, state, chains).reshape(R, Kft, 4, 3)
    logp = logp + torch.from_numpy(rs.normal(0, 0.02, size=logp.shape).astype(np.float32))
    adv = torch.from_numpy(rs.normal(0.5, 2.0, R).astype(np.float32))
    ret = torch.from_numpy(rs.normal(0, 1, R).astype(np.float32))
    val = torch.from_numpy(rs.normal(0, 1, R).astype(np.float32))
    inds = torch.from_numpy(rs.permutation(R * Kft)[:128].astype(np.int64))
    return a, c, cfg, (base, ft, cr), (state, chains, logp, adv, ret, val), inds


def loss_and_grads(a, c, cfg, params, data, inds, global_moments=None):
    base, ft, cr = params
    ft = {k: v.clone().requires_grad_(True) for k, v in ft.items()}
    cr = {k: v.clone().requires_grad_(True) for k, v in cr.items()}
    state, chains, logp, adv, ret, val = data
    Kft = cfg.ft_denoising_steps
    b, k = inds // Kft, inds % Kft
    res = O.ppo_loss(cfg, a, c, base, ft, cr, state[b], chains[b, k], chains[b, k + 1], k, ret[b], val[b], adv[b],
                     logp[b, k], global_moments=global_moments)
    (res[0] + res[2]).backward()
    flat = torch.cat([p.grad.reshape(-1) for p in list(ft.values()) + list(cr.values())])
    stats = torch.tensor([res[0].item(), res[2].item(), res[4], res[3], res[5]], dtype=torch.float64)
    return flat, stats


def worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from dppo_amd.parallel import allreduce_bucket, pool_minibatch_moments
    a, c, cfg, params, data, inds = make_problem()
    mine = inds[rank::world].contiguous()  # this rank's share of the global minibatch
    mom = pool_minibatch_moments(data[3], [mine], cfg.ft_denoising_steps)[0]
    flat, stats = loss_and_grads(a, c, cfg, params, data, mine, global_moments=mom.tolist())
    bucket = torch.cat([flat, stats.float()])
    allreduce_bucket(bucket)
    if rank == 0:
        out.put((mom.numpy(), bucket.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_update_equals_single_process():
    world, port = 2, free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    mom, bucket = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a, c, cfg, params, data, inds = make_problem()
    adv_g = data[3][inds // 10].double()
    np.testing.assert_allclose(mom, [adv_g.sum().item(), (adv_g * adv_g).sum().item(), 128.0], rtol=1e-12)
    flat, stats = loss_and_grads(a, c, cfg, params, data, inds)
    n = flat.numel()
    np.testing.assert_allclose(bucket[n:], stats.numpy(), rtol=2e-5, atol=1e-6)
    g = bucket[:n]
    assert np.linalg.norm(g - flat.numpy()) <= 1e-4 * np.linalg.norm(flat.numpy())


def test_moments_single_process_passthrough():
    from dppo_amd.parallel import pool_minibatch_moments
    adv = torch.arange(12, dtype=torch.float32)
    m = pool_minibatch_moments(adv, [torch.tensor([0, 5, 23, 47])], 4)  # rows 0, 1, 5, 11
    np.testing.assert_allclose(m.numpy(), [[17.0, 0 + 1 + 25 + 121, 4.0]])
