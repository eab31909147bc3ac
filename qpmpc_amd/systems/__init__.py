"""System models used as problem generators (reference: qpmpc/systems/)."""
from .wheeled_inverted_pendulum import WheeledInvertedPendulum  # noqa: F401
