// placeholder, replaced below
