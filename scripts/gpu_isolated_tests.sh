#!/bin/bash
# every test file in its own process (fresh library state: no cache warmed by an earlier file), then the suite in reverse file order
for f in tests/test_*.py; do
  r=$(timeout 900 python -m pytest $f -m gpu -q --tb=line -p no:cacheprovider 2>&1 | tail -1)
  echo "$f: $r"
done
echo "== reverse file order, one process"
timeout 1500 python -m pytest $(ls tests/test_*.py | sort -r) -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -4
