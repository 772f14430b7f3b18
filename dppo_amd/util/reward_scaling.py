"""Running reward scaling on the host (numpy float64), stateful across iterations.

Mirrors ``dppo/util/reward_scaling.py:13-87`` (reference): rewards are divided by the running std of the
forward-discounted return (pooled over envs; note the reference's ``M2 / (tot_count - 1)``) and clipped to +-10.
``moments_hook`` lets data-parallel runs pool the batch moments over ranks before the update.
"""
import numpy as np


class RunningMeanStd:
    def __init__(self, epsilon=1e-4, shape=()):
        self.mean = np.zeros(shape)
        self.var = np.ones(shape)
        self.count = epsilon

    def update(self, x):
        self.update_from_moments(np.mean(x, axis=0), np.var(x, axis=0), x.shape[0])

    def update_from_moments(self, batch_mean, batch_var, batch_count):
        delta = batch_mean - self.mean
        tot = self.count + batch_count
        self.mean = self.mean + delta * batch_count / tot
        m2 = self.var * self.count + batch_var * batch_count + delta ** 2 * self.count * batch_count / tot
        self.var = m2 / (tot - 1)
        self.count = tot


def backward_discounted_sum(prevret, reward, first, gamma):
    assert first.ndim == 2
    ret = np.zeros_like(reward)
    for t in range(reward.shape[1]):
        prevret = ret[:, t] = reward[:, t] + (1 - first[:, t]) * gamma * prevret
    return ret


class RunningRewardScaler:
    def __init__(self, num_envs, cliprew=10.0, gamma=0.99, epsilon=1e-8, per_env=False, moments_hook=None):
        self.ret_rms = RunningMeanStd(shape=(num_envs,) if per_env else ())
        self.cliprew, self.gamma, self.epsilon, self.per_env = cliprew, gamma, epsilon, per_env
        self.ret = np.zeros(num_envs)
        self.moments_hook = moments_hook  # (mean, var, count) -> pooled (mean, var, count)

    def __call__(self, reward, first):
        rets = backward_discounted_sum(self.ret, reward, first, self.gamma)
        self.ret = rets[:, -1]
        x = rets if self.per_env else rets.reshape(-1)
        mean, var, cnt = np.mean(x, axis=0), np.var(x, axis=0), x.shape[0]
        if self.moments_hook is not None:
            mean, var, cnt = self.moments_hook(mean, var, cnt)
        self.ret_rms.update_from_moments(mean, var, cnt)
        return self.transform(reward)

    def transform(self, reward):
        return np.clip(reward / np.sqrt(self.ret_rms.var + self.epsilon), -self.cliprew, self.cliprew)
