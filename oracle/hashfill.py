"""TEST INFRASTRUCTURE — the hash generator moved to ``workloads.hashfill`` (shared with the benches); re-exported here."""
from workloads.hashfill import *       # noqa: F401,F403
from workloads.hashfill import HashedNoise, fill_state_dict, normal, uniform, uniform01       # noqa: F401
