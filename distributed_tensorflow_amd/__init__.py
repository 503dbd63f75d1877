"""MI355X-native parameter-server training framework (see README.md)."""
__version__ = "0.1.0"
