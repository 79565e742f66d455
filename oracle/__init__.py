"""ORACLE -- CPU restatement of the reference's hot path.  Test infrastructure only: see ops_ref.py."""
