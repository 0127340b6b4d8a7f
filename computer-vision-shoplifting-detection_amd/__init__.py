"""MI355X-native YOLO detect/pose hot path (import it through the ``cvsd_amd`` alias package)."""
