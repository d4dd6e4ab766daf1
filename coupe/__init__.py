"""Namespace package for the MI355X-native coupe.DVSG hot path (see coupe.dvsg_amd)."""
