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


# ------------------------------------------------------------------ the DataParallel class itself, world_size 2
def dp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from dppo_amd.agent.finetune.train_ppo_diffusion_agent import TrainPPODiffusionAgent
    from dppo_amd.model.common.critic import CriticObs
    from dppo_amd.model.diffusion.diffusion_ppo import PPODiffusion
    from dppo_amd.model.diffusion.mlp_diffusion import DiffusionMLP
    from dppo_amd.parallel import STATS_SLOTS, DataParallel
    a, c, cfg, params, data, inds = make_problem()
    # every rank starts from DIFFERENT weights: the constructor's broadcast must make them rank 0's
    actor = DiffusionMLP(3, 4, 11, mlp_dims=[512, 512, 512], activation_type="ReLU", residual_style=True)
    critic = CriticObs(11, [256, 256, 256], residual_style=True)
    actor.load_state_dict(O.init_params(a, 100 + rank))
    critic.load_state_dict(O.init_params(c, 200 + rank))
    model = PPODiffusion(actor=actor, critic=critic, horizon_steps=4, obs_dim=11, action_dim=3, device="cpu", **KW)
    model.actor_ft.load_state_dict(O.init_params(a, 300 + rank))
    epochs = [model.actor_ft._epoch, model.critic._epoch, model.actor._epoch]
    dp = DataParallel(model, world)
    res = {}
    # (1) bucket aliasing: the networks' flat gradient buffers ARE slices of the one bucket
    na, nc = model.actor_ft.flat_params().numel(), model.critic.flat_params().numel()
    # (layout [critic | actor | stats]: the critic slice is sent first, from inside the library call; the rest is contiguous)
    res["alias"] = (model.critic.flat_grads().data_ptr() == dp.bucket.data_ptr()
                    and model.actor_ft.flat_grads().data_ptr() == dp.bucket.data_ptr() + 4 * nc
                    and dp.bucket.numel() == na + nc + 2 * STATS_SLOTS)
    # (2) broadcast: every rank now holds rank 0's weights, and the kernel images were invalidated
    res["weights"] = [model.actor_ft.flat_params().clone(), model.critic.flat_params().clone(),
                      model.actor.flat_params().clone()]
    res["epochs_bumped"] = [model.actor_ft._epoch > epochs[0], model.critic._epoch > epochs[1], model.actor._epoch > epochs[2]]
    # (3) one update: per-rank gradients (oracle as the per-rank compute, on rank 0's weights) written into the aliased
    # buffers, per-rank float64 statistics, then the class's all-reduce (non-CUDA branch: hi/lo split of the statistics)
    w = (O.init_params(a, 61), O.init_params(a, 300), O.init_params(c, 200))
    mine = inds[rank::world].contiguous()
    mom = dp.minibatch_moments(data[3], [mine], cfg.ft_denoising_steps)[0]
    flat, stats = loss_and_grads(a, c, cfg, w, data, mine, global_moments=mom.tolist())
    model.actor_ft.flat_grads().copy_(flat[:na])
    model.critic.flat_grads().copy_(flat[na:])
    st = torch.zeros(STATS_SLOTS, dtype=torch.float64)
    st[:5] = torch.tensor([stats[0], stats[1], stats[2], stats[3], stats[4]])  # pg, v, kl, clipfrac, ratio partial sums
    st[5], st[6] = 0.123456789012345, 1.987654321098765  # adv mean / std: global values every rank writes
    st_local = st.clone()
    object.__setattr__(model, "_stats", st)
    local = dp.bucket.clone()
    # the product's order of events: the critic slice from the library's callback (stream handle 0 here: CPU tensors), then
    # the rest behind the actor's gradients
    assert dp.critic_hook is not None
    dp.critic_hook(0)
    res["critic_sent_early"] = bool(dp._critic_done)
    dp.allreduce_grads()
    res["grads"] = torch.cat([model.actor_ft.flat_grads(), model.critic.flat_grads()]).clone()
    res["stats"] = model._stats.clone()
    # ... and the single whole-bucket all-reduce on the same local values gives the same bits (split=False never hooks)
    split_bucket = dp.bucket.clone()
    dp.bucket.copy_(local)
    object.__setattr__(model, "_stats", st_local.clone())
    dp.split = False
    assert dp.critic_hook is None
    dp.allreduce_grads()
    res["split_equals_single"] = bool(torch.equal(dp.bucket[:na + nc], split_bucket[:na + nc])
                                      and torch.equal(model._stats, torch.from_numpy(res["stats"].numpy())))
    dp.split = True
    res["moments"] = mom.clone()
    # (4) after the broadcast every rank draws its own random stream (sampler key, permutations)
    TrainPPODiffusionAgent.reseed(42 + rank)
    res["sampler_key"] = int(torch.randint(0, 2 ** 62, (1,)).item())  # what VPGDiffusion.forward draws
    # numpy arrays travel through the queue by value (torch tensors go through shared-memory handles that die with the worker)
    res = {k: ([t.numpy() for t in v] if isinstance(v, list) and torch.is_tensor(v[0]) else v.numpy() if torch.is_tensor(v) else v)
           for k, v in res.items()}
    out.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_class_on_two_ranks():
    world, port = 2, free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a, c, cfg, params, data, inds = make_problem()
    r0, r1 = got[0], got[1]
    assert r0["alias"] and r1["alias"]
    assert r0["critic_sent_early"] and r1["critic_sent_early"]
    assert r0["split_equals_single"] and r1["split_equals_single"]
    assert all(r0["epochs_bumped"]) and all(r1["epochs_bumped"])
    for x, y, spec, seed in zip(r0["weights"], r1["weights"], (a, c, a), (300, 200, 100)):
        assert np.array_equal(x, y)
        want = torch.cat([O.init_params(spec, seed)[n].reshape(-1) for n, _, _ in O.param_shapes(spec)]).numpy()
        assert np.array_equal(x, want)  # rank 0's
    # the reduced gradient / statistics equal the single-process ones on the whole minibatch, on both ranks
    w = (O.init_params(a, 61), O.init_params(a, 300), O.init_params(c, 200))
    flat, stats = loss_and_grads(a, c, cfg, w, data, inds)
    for r in (r0, r1):
        assert np.array_equal(r["grads"], r0["grads"])
        assert np.linalg.norm(r["grads"] - flat.numpy()) <= 1e-4 * flat.norm().item()
        np.testing.assert_allclose(r["stats"][:5], stats.numpy(), rtol=2e-5, atol=1e-6)
        # float64 statistics survive the fp32 bucket (hi + lo), and the two global values are not doubled
        assert float(r["stats"][5]) == pytest.approx(0.123456789012345, rel=1e-13)
        assert float(r["stats"][6]) == pytest.approx(1.987654321098765, rel=1e-13)
    adv_g = data[3][inds // 10].double()
    np.testing.assert_allclose(r0["moments"], [adv_g.sum().item(), (adv_g * adv_g).sum().item(), 128.0], rtol=1e-12)
    assert r0["sampler_key"] != r1["sampler_key"]


# ------------------------------------------------------------------ statistics of a Gaussian head under data parallelism
class _StubNet:
    def __init__(self, n, fill):
        self._p = torch.full((n,), float(fill))
        self._flat_grad = torch.zeros(n)

    def flat_params(self):
        return self._p

    def flat_grads(self):
        return self._flat_grad

    def mark_updated(self):
        pass


class _StubModel:
    def __init__(self, rank, dp_avg_stats=None):
        self.actor_ft, self.critic, self.actor = _StubNet(40, rank), _StubNet(24, rank), _StubNet(40, rank)
        self._stats = None
        if dp_avg_stats is not None:
            self.dp_avg_stats = dp_avg_stats


def gauss_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dppo_amd.model.rl.gaussian_ppo import PPO_Gaussian
    from dppo_amd.parallel import DataParallel
    res = {"class_value": PPO_Gaussian.dp_avg_stats}
    for name, n_avg in (("gaussian", PPO_Gaussian.dp_avg_stats), ("diffusion", None)):
        m = _StubModel(rank, n_avg)
        dp = DataParallel(m, world)
        # 9 slots like hip.GAUSS_STAT_COUNT: [pg, v, kl, clipfrac, ratio] partial sums, [adv mean, adv std, entropy] written in
        # full by every rank (the entropy of a state-independent Gaussian depends on sigma only), [mean std] never travels
        st = torch.tensor([0.1 * (rank + 1), 0.2 * (rank + 1), 0.01, 0.02, 0.5, 0.25, 1.5, -3.75, 0.1], dtype=torch.float64)
        m._stats = st.clone()
        m.critic.flat_grads().fill_(rank + 1.0)
        dp.critic_hook(0)
        dp.allreduce_grads()
        res[name] = m._stats.numpy().copy()
        res[name + "_cgrad"] = float(m.critic.flat_grads()[0])
    out.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_gaussian_entropy_is_not_summed_over_ranks():
    """DPPO_GAUSS_STAT_ENTROPY (slot 7) is a rank-global value; summed over ranks the logged entropy (and the loss
    metric built from it) came out world times too large.  PPO_Gaussian.dp_avg_stats = 3 widens the averaged range."""
    world, port = 2, free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=gauss_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        g, d = got[r]["gaussian"], got[r]["diffusion"]
        assert got[r]["class_value"] == 3
        np.testing.assert_allclose(g[:5], [0.3, 0.6, 0.02, 0.04, 1.0], rtol=2e-7)   # partial sums: summed (in the fp32 bucket)
        np.testing.assert_allclose(g[5:8], [0.25, 1.5, -3.75], rtol=1e-12)           # global values: not doubled
        assert g[8] == 0.1                                                            # never travelled
        np.testing.assert_allclose(d[5:7], [0.25, 1.5], rtol=1e-12)
        assert d[7] == pytest.approx(-7.5, rel=1e-6)  # a diffusion model's slot 7 is an ordinary partial sum (unused there)
        assert got[r]["gaussian_cgrad"] == 3.0
