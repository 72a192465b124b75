"""MI355X-native ORB front end: drop-in for ORB-SLAM2's ORBextractor / ORBmatcher hot path."""
from .bow import ORBVocabulary  # noqa: F401
from .extractor import ORBextractor  # noqa: F401
from .frame import Frame, KeyFrame, MapPoint, compute_distinctive_descriptors  # noqa: F401
from .matcher import ORBmatcher, knn2_bruteforce, stereo_match  # noqa: F401
