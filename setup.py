"""`pip install .` puts the package and the two top-level drop-in modules the reference's
wrappers import (`hgnnaggr`, `unignnaggr`; reference setup.py:18,32-33) on the import path.
The HIP library is built in-tree by `make -C hypergef_amd/csrc` (hipcc, gfx950) and shipped as
package data; there is no CPU build."""
import os
import subprocess

from setuptools import setup
from setuptools.command.build_py import build_py

ROOT = os.path.dirname(os.path.abspath(__file__))


class BuildWithHip(build_py):
    def run(self):
        subprocess.run(["make", "-C", os.path.join(ROOT, "hypergef_amd", "csrc")], check=True)
        super().run()


setup(
    name="hypergef_amd",
    version="0.2.0",
    description="MI355X-native fused hypergraph aggregation behind the HyperGef operator surface",
    packages=["hypergef_amd"],
    py_modules=["hgnnaggr", "unignnaggr"],
    package_data={"hypergef_amd": ["lib/*.so"]},
    cmdclass={"build_py": BuildWithHip},
)
