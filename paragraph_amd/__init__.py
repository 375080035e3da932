"""paragraph_amd -- MI355X-native read -> variant-graph realignment core (drop-in for the
grm::alignReads / GraphAligner path of Illumina/paragraph).  See DESIGN.md."""
from . import capi  # noqa: F401
from .capi import AF_ALL, AF_BOTH_STRANDS, AF_CIGAR, AF_REVERSE_GRAPH, Context, PgError  # noqa: F401

__all__ = ["capi", "Context", "PgError", "AF_ALL", "AF_BOTH_STRANDS", "AF_CIGAR", "AF_REVERSE_GRAPH"]
