"""smartstartcontinuous_amd -- MI355X-native rollout engine behind the gym.Env / RLAgent
surface of darren-huang/SmartStartContinuous.

Importing the package does not touch the GPU; the HIP library (libssc.so) is loaded on
first use and its absence is an error -- there is no CPU fallback."""

__version__ = "0.1.0"

from . import _ffi  # noqa: F401
from .spaces import Box, EnvSpec  # noqa: F401
from .vec_env import (ActorPolicy, Continuous_MountainCarEnv_Editted, EpisodeRing, MpcPolicy, RandomPolicy,  # noqa: F401
                      SingleEnvView, TransitionChunk, VecEnv, make)
from .rl_train import Episode, Summary, rlTrain, rl_train_vec, rl_train_vec_ddpg, rl_train_vec_smartstart  # noqa: F401,E402
from .replay_buffer import DeviceReplayBuffer, ReplayBuffer  # noqa: F401,E402
from .smartstart import SmartStartContinuous, VecSmartStart  # noqa: F401,E402
from .collect_samples import (CollectSamples, Policy_Random, TrainingSet, dataset_from_chunk,  # noqa: F401,E402
                              generate_training_data_inputs, generate_training_data_outputs, perform_rollouts)
