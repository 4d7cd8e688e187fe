"""nubovca -- python harness around libnubovca_hip (MI355X Haar detection hot path)."""
