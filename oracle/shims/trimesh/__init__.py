"""Inert stub: lets `dust3r.viz` import in the build container. No functionality."""
