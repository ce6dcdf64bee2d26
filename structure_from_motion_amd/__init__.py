"""MI355X-native bundle adjustment behind the reference's ``BundleAdjustment`` interface."""
__version__ = "0.1.0"
