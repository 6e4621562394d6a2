"""Measurement-based GKP layer on the MI355X CV backend (the surface of ``simulators/gkp_simulator``).

Host bookkeeping only -- logical gates compile to ``InsertBell`` / ``BS`` / ``Homodyne`` sequences, homodyne outcomes are
folded into a Pauli frame -- on top of ``cv_simulator`` with ``MPS(layout="sites")``, where the work happens.
"""
