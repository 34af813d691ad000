#!/bin/bash
# Builds opus_amd/_build/libopusfe_diag.so with -DOPUSFE_DIAG: the diagnostic switch of run_lower (OPUSFE_LOWER_SEQ: every function through
# the sequential machine) exists only in this build; load it
# through OPUSFE_LIB.  The shipped libopusfe.so reads none of them.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread -DOPUSFE_DIAG -I $ROOT/include -shared -o $ROOT/opus_amd/_build/libopusfe_diag.so $ROOT/opus_amd/csrc/api.hip
echo $ROOT/opus_amd/_build/libopusfe_diag.so
