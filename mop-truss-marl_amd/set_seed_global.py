seedThis = 20
