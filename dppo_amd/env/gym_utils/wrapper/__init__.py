"""Per-environment wrappers of the reference's host env stack (env/gym_utils/wrapper): names as in its ``wrapper_dict``."""
from dppo_amd.env.gym_utils.wrapper.mujoco_locomotion_lowdim import MujocoLocomotionLowdimWrapper
from dppo_amd.env.gym_utils.wrapper.multi_step import MultiStep

wrapper_dict = {"mujoco_locomotion_lowdim": MujocoLocomotionLowdimWrapper, "multi_step": MultiStep}
