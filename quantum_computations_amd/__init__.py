"""quantum_computations_amd -- MI355X-native gate application behind the reference's ``simulators`` API.

``dv_simulator`` / ``cv_simulator`` mirror the reference packages of the same names; ``device`` holds the
HBM-resident registers; ``_lib`` binds libqsv.so (include/qsv.h).  Importing the package does not touch the
GPU; the first register does, and raises if libqsv.so has not been built.
"""
__version__ = "0.1.0"
