#!/bin/bash
# AddressSanitizer + UBSan over the CPU-side C code (the NetCDF classic I/O library and the test oracle), driven by
# their pytest files.  GPU sanitizers are not available on the pool, so this is the sanitizer coverage of the build.
# The in-tree .so files are replaced by instrumented ones for the run and rebuilt normally afterwards.
set -euo pipefail
cd "$(dirname "$0")/.."
SAN="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer"
PRE="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
restore() {
  python -c "from mpassit_amd import build; build.build_ncio(force=True)"
  make -C oracle clean >/dev/null && make -C oracle >/dev/null
}
trap restore EXIT
H5=$(python -c "from mpassit_amd import build; h = build.find_hdf5(); print('-DMPASSIT_HAVE_HDF5 -I%s %s/libhdf5_hl.so %s/libhdf5.so -Wl,--enable-new-dtags,-rpath,%s' % (h[0], h[1], h[1], h[1]) if h else '')")
gcc $SAN -shared -fPIC -pthread -o mpassit_amd/hostio/libmpassit_ncio.so mpassit_amd/hostio/ncclassic.c $H5   # the NetCDF-4 backend too where libhdf5 exists
make -C oracle clean >/dev/null
make -C oracle CFLAGS="$SAN -fPIC -fopenmp -ffp-contract=off" >/dev/null
LD_PRELOAD="$PRE" ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python -m pytest tests/test_ncio.py tests/test_nc4.py tests/test_ncio_property.py tests/test_target_grid_file.py tests/test_oracle.py \
    tests/test_projection_properties.py tests/test_projection_goldens.py tests/test_weight_goldens.py tests/test_store_goldens.py tests/test_esmf_pin.py -x -q -m "not gpu"
