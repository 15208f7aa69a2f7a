"""MI355X-native constraint-evaluation engine for Gatekeeper's hot path (Match + Rego-driver Query).

Only what the path needs: csrc/ (HIP kernels + C ABI, built into libgkgpu.so), _lib.py (ctypes binding),
driver.py (host mirror of drivers.Driver / Client.Review), synth.py (synthetic workload generator).
"""
