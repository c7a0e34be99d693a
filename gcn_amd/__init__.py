"""gcn_amd -- the GCN propagation path A_hat.(X.W) of hareeshks/gcn as HIP kernels for MI355X.

``gcn_amd.sparse`` holds the device CSR type and the operators; ``gcn_amd.utils``,
``layers``, ``models``, ``metrics`` and ``inits`` mirror the reference's modules of the
same names.
"""
__version__ = '0.1.0'
