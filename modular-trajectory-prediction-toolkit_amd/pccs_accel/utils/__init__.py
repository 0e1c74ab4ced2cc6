"""Modules named like the reference's ``utils`` package (PCCSNet/utils)."""
