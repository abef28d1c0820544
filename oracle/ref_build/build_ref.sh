#!/usr/bin/env bash
# Builds oracle/_ref/libref_rasterizer.so: the REFERENCE rasterizer (INRIA diff-gaussian-rasterization as vendored by
# SuGaR) compiled by hipcc for gfx950 from its sources where they lie under /root/reference.  TEST INFRASTRUCTURE ONLY
# (parity checker and A/B timing baseline on the GPU box); never linked into or loaded by the product.
#
# The sources are used unmodified except that nvcc's whitespace-tolerant launch token "<< <...>> >" is respelled
# "<<<...>>>" on the fly (clang requires the contiguous token); the respelled stream lives in a temp dir that is deleted,
# nothing from the reference is copied into the repository.  CUDA-only headers resolve to the shims in ./shim.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
DGR=/root/reference/gaussian_splatting/submodules/diff-gaussian-rasterization
KNN=/root/reference/gaussian_splatting/submodules/simple-knn
OUT="$HERE/../_ref"
[ -d "$DGR" ] || { echo "[build_ref] $DGR not present (GPU box): using the prebuilt $OUT if any"; exit 0; }
# The reference's own Python for the path (SuGaR model + vanilla 3DGS model / renderer / losses), staged UNMODIFIED next to the
# libraries so that the -m gpu tests and `bench.py --reference-loop` can run the reference's classes on the GPU box, where
# /root/reference does not exist (oracle/_ref/ is git-ignored: nothing of it enters the repository's history).
stage_py() {
  local dst="$OUT/pysrc"
  mkdir -p "$dst/gaussian_splatting"
  for d in sugar_scene sugar_utils sugar_trainers sugar_extractors; do
    mkdir -p "$dst/$d"; cp -f /root/reference/$d/*.py "$dst/$d/"
  done
  for d in gaussian_renderer scene utils arguments; do
    mkdir -p "$dst/gaussian_splatting/$d"; cp -f /root/reference/gaussian_splatting/$d/*.py "$dst/gaussian_splatting/$d/"
  done
  cp -f /root/reference/gaussian_splatting/train.py "$dst/gaussian_splatting/train.py"
}
mkdir -p "$OUT"; stage_py
STAMP="$OUT/.stamp"
SIG="$(cat "$DGR"/cuda_rasterizer/*.cu "$DGR"/cuda_rasterizer/*.h "$KNN"/simple_knn.cu "$KNN"/simple_knn.h "$HERE"/ref_capi.cpp "$HERE"/ref_knn_capi.cpp "$HERE"/shim/*.h "$HERE"/build_ref.sh | sha256sum | cut -d' ' -f1)"
if [ -f "$OUT/libref_rasterizer.so" ] && [ -f "$OUT/libref_simple_knn.so" ] && [ -f "$STAMP" ] && [ "$(cat "$STAMP")" = "$SIG" ]; then exit 0; fi
mkdir -p "$OUT"
TMP="$(mktemp -d)"; trap 'rm -rf "$TMP"' EXIT
FLAGS="-x hip -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w -I$HERE/shim -I$DGR/third_party/glm -I$DGR/cuda_rasterizer"
# Two builds of the same sources:
#   libref_rasterizer.so            compiler defaults (FMA contraction on, like nvcc's default -fmad=true): the A/B baseline
#   libref_rasterizer_nocontract.so -ffp-contract=off: every float op individually rounded, i.e. the arithmetic contract of
#                                   oracle/cpu_rasterizer.c and of the product's preprocess kernels -> compared BIT-EXACTLY
pids=()
for f in forward backward rasterizer_impl; do
  sed -E 's/<<[[:space:]]+</<<</g; s/>>[[:space:]]+>/>>>/g' "$DGR/cuda_rasterizer/$f.cu" > "$TMP/$f.cu"
  hipcc $FLAGS -c "$TMP/$f.cu" -o "$TMP/$f.o" & pids+=($!)
  hipcc $FLAGS -ffp-contract=off -c "$TMP/$f.cu" -o "$TMP/${f}_nc.o" & pids+=($!)
done
hipcc $FLAGS -c "$HERE/ref_capi.cpp" -o "$TMP/ref_capi.o" & pids+=($!)
# simple-knn (distCUDA2): the reference's simple_knn.cu, individually rounded float ops like the C restatement it pins
# (oracle/cpu_rasterizer.c: oracle_dist2).  <cfloat> is force-included: the source uses FLT_MAX without including it
# (it compiled with the CUDA 11.8 headers the reference pins, environment.yml).
sed -E 's/<<[[:space:]]+</<<</g; s/>>[[:space:]]+>/>>>/g' "$KNN/simple_knn.cu" > "$TMP/simple_knn.cu"
hipcc $FLAGS -ffp-contract=off -include cfloat -I"$KNN" -c "$TMP/simple_knn.cu" -o "$TMP/simple_knn.o" & pids+=($!)
hipcc $FLAGS -I"$KNN" -c "$HERE/ref_knn_capi.cpp" -o "$TMP/ref_knn_capi.o" & pids+=($!)
for p in "${pids[@]}"; do wait "$p"; done
hipcc -shared -fPIC --offload-arch=gfx950 -o "$OUT/libref_rasterizer.so" "$TMP"/forward.o "$TMP"/backward.o "$TMP"/rasterizer_impl.o "$TMP"/ref_capi.o
hipcc -shared -fPIC --offload-arch=gfx950 -o "$OUT/libref_rasterizer_nocontract.so" "$TMP"/forward_nc.o "$TMP"/backward_nc.o "$TMP"/rasterizer_impl_nc.o "$TMP"/ref_capi.o
hipcc -shared -fPIC --offload-arch=gfx950 -o "$OUT/libref_simple_knn.so" "$TMP"/simple_knn.o "$TMP"/ref_knn_capi.o
echo "$SIG" > "$STAMP"
echo "[build_ref] built $OUT/libref_rasterizer.so, libref_rasterizer_nocontract.so and libref_simple_knn.so"
