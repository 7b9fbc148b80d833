"""Event-stream readers and windowing in front of the hot path (reference: data_readers/)."""
from .event_readers import FixedSizeEventReader, RefTimeEventReaderZip, SingleEventReaderNpz   # noqa: F401
from .video_readers import VR, read_timestamps_file                                              # noqa: F401
from .MVSEC import MVSEC_NE                                                                      # noqa: F401
