"""dppo_amd -- MI355X-native DPPO hot path (K-step diffusion sampler + PPO update) behind the
reference's module surface.  Module paths mirror ``dppo.*`` so a Hydra ``_target_`` only needs the
package prefix changed.  Compute lives in ``lib/libdppo_hip.so`` (``include/dppo_hip.h``).
"""
__version__ = "0.1.0"
