"""MI355X-native ORB front end: drop-in for ORB-SLAM2's ORBextractor / ORBmatcher hot path."""
from .extractor import ORBextractor  # noqa: F401
