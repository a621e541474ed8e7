for rep in 1 2 3; do for v in 0 1; do
  EAGLE_CONV_48NR=$v python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('48NR=$v', j['value'], 'fps  conv', j['roofline']['conv_ms_per_step'], 'ms/step', [(r['layer'],r['launches_per_step'],r['avg_us']) for r in j['roofline_conv_layers'] if '48->48 @135' in r['layer']])"
done; done
