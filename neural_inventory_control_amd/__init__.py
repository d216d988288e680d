"""MI355X-native differentiable inventory-rollout engine (drop-in for the hot path of Neural_inventory_control)."""
__version__ = "0.1.0"
