"""
ntjoin_amd -- MI355X-native minimizer-sketch + minimizer-graph engine for ntJoin's hot path.

The compute lives in hand-written HIP kernels (ntjoin_amd/csrc) behind the C-ABI declared in
include/ntjoin_mx.h; this package is the thin Python host side:

  ntjoin_amd.capi          ctypes binding of libntjoin_mx.so (fails loudly if the library is missing)
  ntjoin_amd.engine        MxEngine: one handle = one (k, w) problem: add assemblies -> sketch -> graph
  ntjoin_amd.ntjoin_utils  drop-in counterparts of the reference's read_minimizers / filter_minimizers /
                           build_graph (reference bin/ntjoin_utils.py:83-193)
  ntjoin_amd.ntjoin        counterpart of the reference's Ntjoin.load_minimizers / make_minimizer_graph /
                           print_graph (reference bin/ntjoin.py:25-67,178-204)
  ntjoin_amd.indexlr       `indexlr`-flag-compatible command line (reference ntJoin:204-205)
  ntjoin_amd.dist          one-process-per-GPU sharding + RCCL all-gather of sketches
"""
__version__ = "0.1.0"
