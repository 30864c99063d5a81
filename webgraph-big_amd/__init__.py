"""webgraph-big_amd — MI355X-native BVGraph successor-list decoding.

The package holds only what the decode path needs: csrc/ (HIP kernels + C ABI), host/ (the C++ mirror), bvgraph.py (host-side
mirror of the reference's ImmutableGraph / NodeIterator / LazyLongIterator API over the C ABI) and shard.py (the node-range
shard helpers bench.py and the tests share); experimental/ holds kernels that lost to the product's and are built only by
`make experimental`.  The CPU encoder and the synthetic graph generators are test tooling and live outside the package (tooling/).
"""
from ._abi import *  # noqa: F401,F403
from ._abi import Params, ScanResult, Tuning, default_params  # noqa: F401
from .bvgraph import (BVGraph, NodeIterator, LazyLongIterator, BVGraphError, IllegalArgumentException,  # noqa: F401
                      IllegalStateException, UnsupportedOperationException, IOException, EOFException, DeviceError,
                      NoSuchElementException, parse_properties, decode_offsets, arc_mix, build, lib, library_path,
                      BitStreamArcLabelledImmutableGraph, LabelledArcIterator, parse_label_spec, LABEL_GAMMA_INT, LABEL_FIXED_INT, LABEL_FIXED_INT_LIST, LABEL_FIXED_LONG_LIST,
                      scan_multi, mosaic, BALANCE_NODES, BALANCE_BITS, BALANCE_ARCS, store)
