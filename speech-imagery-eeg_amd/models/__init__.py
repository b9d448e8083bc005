"""Model registry of the drop-in (keys follow IGN/exp/experiment_classification.py:87-93 and
IGN/model/InterpGN.py:13-19).  The reference imports this package as ``models`` although its directory
is called ``model``; ``model/`` here is an alias of this package."""
