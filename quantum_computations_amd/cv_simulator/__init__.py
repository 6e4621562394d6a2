"""MI355X-backed drop-in for the gate-application path of ``simulators.cv_simulator`` (position-grid CV circuits)."""
import logging

# Logger names follow the reference's module paths ("simulators.cv_simulator.gates", ...): its scripts configure and
# silence logging.getLogger("simulators") (impact_.../grover.py:24), and that must act on this package too.
logging.getLogger("simulators." + __name__.split(".", 1)[1]).addHandler(logging.NullHandler())
