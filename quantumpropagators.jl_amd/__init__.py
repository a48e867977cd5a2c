"""MI355X-native `prop_step!` engine for QuantumPropagators.jl (import name ``qprop_amd``).

Only what the hot path needs lives here: ``csrc/`` (HIP kernels + the C ABI declared
in ``include/qprop.h``), ``lib.py`` (ctypes binding, loads ``libqprop_hip.so`` and fails
loudly when it is missing), ``propagator.py`` (host-side mirror of the reference's
``init_prop / prop_step! / reinit_prop! / set_state! / set_t! / propagate`` interface),
``sharded.py`` (row-partitioned multi-GPU driver over torch.distributed) and
``synth.py`` (synthetic inputs).
"""
__version__ = "0.1.0"
