"""ice_halo_sim_amd — MI355X-native trace backend for the Lumice ice-halo simulator (one hot path only)."""
__version__ = "0.1.0"
