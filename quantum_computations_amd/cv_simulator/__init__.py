"""MI355X-backed drop-in for the gate-application path of ``simulators.cv_simulator`` (position-grid CV circuits)."""
import logging

logging.getLogger(__name__).addHandler(logging.NullHandler())
