"""The boundary is a C ABI: include/edigpu.h must compile as plain C (C99) and as C++, and a C program must
link against libedigpu.so using nothing but that header.  CPU only (the program is not run)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

C_SRC = r"""
#include <stdio.h>
#include "edigpu.h"
int main(void) {
  struct edigpu_model m;
  edigpu_handle h = 0;
  int64_t dim = 0;
  int n = 0;
  double ev[2];
  (void)sizeof(m);
  if (edigpu_device_count(&n) != 0) { printf("%s\n", edigpu_last_error()); return 0; }
  if (edigpu_normal_build(&h, &m, 1, 1, 0, -1) == 0) {
    edigpu_sector_dim(&m, 1, 1, &dim);
    edigpu_lanczos_eigh_multi(h, 2, 0, 1e-10, 0, 0, ev, 0, 0, 0);
    edigpu_destroy(h);
  }
  return edigpu_version() > 0 ? 0 : 1;
}
"""


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_header_is_c99_and_links(built, tmp_path):
    src = tmp_path / "use_edigpu.c"
    src.write_text(C_SRC)
    inc = os.path.join(ROOT, "include")
    lib = os.path.join(ROOT, "edipack_amd", "lib")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", inc, "-c", str(src),
                           "-o", str(tmp_path / "use_edigpu.o")])
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I", inc, "-x", "c++", "-c", str(src),
                           "-o", str(tmp_path / "use_edigpu_cpp.o")])
    # link only (running needs a GPU): every symbol the program uses must be exported by the library
    subprocess.check_call(["gcc", str(tmp_path / "use_edigpu.o"), "-L", lib, "-ledigpu", "-Wl,-rpath," + lib,
                           "-Wl,--allow-shlib-undefined", "-o", str(tmp_path / "use_edigpu")])
