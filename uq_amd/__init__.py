"""uq_amd -- MI355X-native (gfx950) encode/decode hot path of the uQ binary-FASTQ format."""
__version__ = '0.1.0'
