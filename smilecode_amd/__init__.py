"""smilecode_amd -- MI355X-native (gfx950, hand-written HIP) drop-in for the ModeT hot path of
ZAX130/SmileCode.  ``from smilecode_amd.models import ModeT`` replaces ``from models import ModeT``."""
__version__ = "0.1.0"
