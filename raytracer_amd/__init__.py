"""MI355X-native path-tracing hot path: host scene pipeline (host), device library binding
(device), record layouts (layouts) and synthetic scenes (procedural)."""
from . import layouts  # noqa: F401

__all__ = ["layouts", "host", "device", "procedural"]
