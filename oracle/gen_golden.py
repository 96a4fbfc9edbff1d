#!/usr/bin/env python3
"""Generate tests/golden/glm_*.f32 from the reference's vendored GLM and tests/golden/ref_* from the reference's own
aabb / HittableList / bvh_node / checker_texture headers (dev container only).

Builds oracle/_ref/glm_probe (oracle/Makefile `ref`: g++ on ref_glm_probe.cpp with
-I/root/reference/Libraries/include -I/root/reference/main/src — the reference sources are
compiled where they lie, never copied) and runs it.  The fixtures are data (inputs + expected
outputs); commit them together with this script.
"""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
out = os.path.join(here, "..", "tests", "golden")
os.makedirs(out, exist_ok=True)
subprocess.check_call(["make", "-C", here, "ref"])
subprocess.check_call([os.path.join(here, "_ref", "glm_probe"), out])
# the reference's headers above the vocabulary (oracle/ref_path_probe.cpp; <cuda_runtime.h> = NVIDIA's own, from the triton wheel)
subprocess.check_call([os.path.join(here, "_ref", "path_probe"), out])
