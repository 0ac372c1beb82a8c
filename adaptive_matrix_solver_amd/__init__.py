"""MI355X-native accelerator for the MAUS per-candidate inner loop."""
from ._cabi import Context, MausHipError, load_library  # noqa: F401
