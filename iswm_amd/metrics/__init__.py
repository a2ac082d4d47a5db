from .stream_metrics import StreamMetrics

__all__ = ["StreamMetrics"]
