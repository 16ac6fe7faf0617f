"""Drop-in mirror of the reference's ``models`` package (models/__init__.py:1):
same module names, class names, constructor arguments and state_dict keys; the
eval-mode ``forward`` runs on the MI355X-native engine (libafx.so)."""
from models.fe import *  # noqa: F401,F403
