"""Builds the REFERENCE's own Cython wrapper (poismf.c_funs_double / poismf.c_funs_float) against the MI355X libraries.

Run it from the root of a checkout of david-cortes/poismf, in place of that project's setup.py:

    POISMF_HIP_LIB_DIR=/path/to/poismf_amd python3 /path/to/scripts/dropin/setup_hip.py build_ext --inplace

What changes against the reference's build (ref setup.py:225-243): nothing in the wrapper's sources.  The Cython file keeps
calling `run_poismf`, `factors_multiple`, `predict_multiple` and `topN` by name (ref poismf/poismf_c_wrapper.pxi:95-103, :192-200,
:112, :241); those four names now resolve into libpoismf_hip_{d,f}.so, which exports them with the prototypes of ref
src/poismf.h:226-233, :270-280, :247-254, :236-244.  The five C sources of the core are STILL compiled -- `factors_single`
(ref src/pred.c:201-307) stays on the CPU and needs `tnc`, `calc_fun_and_grad`, and pred.c's own `factors_multiple` body needs
`pg_iteration` / `cg_iteration` / `tncg_iteration` (ref src/pred.c:163-182) -- but with the four entry points renamed by the
preprocessor (`-Drun_poismf=run_poismf_cpu` ...), for these five files only, so that the wrapper's calls are left undefined
in the objects and bind to the HIP library at load time.  The renamed CPU bodies stay in the module as `*_cpu`.

Per-file macros are why the C core is compiled here by hand into objects (one set per precision) instead of being listed in
`Extension(sources=...)`: `define_macros` of an Extension would rename the calls in the Cython-generated file as well.
"""
import os
import subprocess
import sys

import numpy
from Cython.Distutils import build_ext
from setuptools import Extension, setup

HIP_LIB_DIR = os.path.abspath(os.environ.get("POISMF_HIP_LIB_DIR", "poismf_amd"))
CORE = ["src/poismf.c", "src/nonnegcg.c", "src/tnc.c", "src/pred.c", "src/topN.c"]
# the entry points the HIP library takes over; the CPU bodies keep existing under the second name
RENAMED = ["run_poismf", "factors_multiple", "predict_multiple", "topN"]
CORE_FLAGS = ["-O3", "-std=c99", "-fopenmp", "-fPIC", "-fno-math-errno", "-fno-trapping-math", "-D_FOR_PYTHON", "-DNDEBUG", "-Isrc"]
CORE_FLAGS += [f"-D{name}={name}_cpu" for name in RENAMED]


class build_ext_hip(build_ext):
    def build_extensions(self):
        cc = os.environ.get("CC", "gcc")
        for ext in self.extensions:
            flavour = "f" if ext.name.endswith("float") else "d"
            out_dir = os.path.join(self.build_temp, "core_" + flavour)
            os.makedirs(out_dir, exist_ok=True)
            for src in CORE:
                obj = os.path.join(out_dir, os.path.basename(src)[:-2] + ".o")
                cmd = [cc, "-c", src, "-o", obj] + CORE_FLAGS + (["-DUSE_FLOAT"] if flavour == "f" else [])
                print(" ".join(cmd))
                subprocess.check_call(cmd)
                ext.extra_objects.append(obj)
        build_ext.build_extensions(self)


def extension(name, pyx, lib, macros):
    return Extension(name, sources=[pyx], include_dirs=[numpy.get_include(), "src/"],
                     define_macros=[("_FOR_PYTHON", None), ("NDEBUG", None)] + macros,
                     libraries=[lib], library_dirs=[HIP_LIB_DIR], runtime_library_dirs=[HIP_LIB_DIR],
                     extra_compile_args=["-O3", "-fopenmp"],
                     extra_link_args=["-fopenmp"])


if not os.path.exists(os.path.join(HIP_LIB_DIR, "libpoismf_hip_d.so")):
    sys.exit(f"setup_hip.py: no libpoismf_hip_d.so under {HIP_LIB_DIR} (set POISMF_HIP_LIB_DIR; python -m poismf_amd.build makes it)")

setup(
    name="poismf",
    packages=["poismf"],
    cmdclass={"build_ext": build_ext_hip},
    ext_modules=[
        extension("poismf.c_funs_double", "poismf/cfuns_double.pyx", "poismf_hip_d", []),
        extension("poismf.c_funs_float", "poismf/cfuns_float.pyx", "poismf_hip_f", [("USE_FLOAT", None)]),
    ],
)
