"""gp_amd — MI355X-native GP training hot path (exact / FITC / VFE NLL+gradient and predict).

The product is gp_amd/libgpx.so (HIP for gfx950 behind the C ABI in include/gpx.h); this package
holds its ctypes binding (gp_amd.lib), the host-side mirror of the reference's class API
(gp_amd.api) and the synthetic-data generator used by the benchmark (gp_amd.synth).
"""
