"""webgraph-big_amd — MI355X-native BVGraph successor-list decoding."""
from ._abi import *  # noqa: F401,F403
